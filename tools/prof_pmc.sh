#!/bin/bash
# rocprofv3 PMC pass (counters only + kernel trace). usage: tools/prof_pmc.sh <tag> "<counters>" [bench args]
TAG=$1; CNT=$2; shift; shift
R=$PWD; mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace -d $R/gpurun_out/pmc_$TAG -o t --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --cpu-seconds 0 --no-roofline "$@" > $R/gpurun_out/pmc_$TAG/bench.log 2>&1
ls $R/gpurun_out/pmc_$TAG
python - <<PY
import csv, collections
f='$R/gpurun_out/pmc_$TAG/t_counter_collection.csv'
rows=list(csv.DictReader(open(f)))
print(rows[0].keys() if rows else 'no rows')
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r['Kernel_Name']
    if 'pdwt' not in n: continue
    key=(n.split('(')[0].replace('void pdwt::',''), r['Grid_Size'])
    d[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items()):
    print(k, {c: '%.3g'%(sorted(x)[len(x)//2]) for c,x in v.items()})
PY
