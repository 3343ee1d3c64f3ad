#!/bin/bash
# rocprofv3 PMC pass (counters only + kernel trace) over an arbitrary command. usage: tools/prof_pmc_cmd.sh <tag> "<counters>" cmd...
TAG=$1; CNT=$2; shift; shift
R=$PWD; mkdir -p $R/gpurun_out/pmc_$TAG
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace -d $R/gpurun_out/pmc_$TAG -o t --output-format csv -- "$@" > $R/gpurun_out/pmc_$TAG/cmd.log 2>&1
python - <<PY
import csv, collections
f='$R/gpurun_out/pmc_$TAG/t_counter_collection.csv'
rows=list(csv.DictReader(open(f)))
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n=r['Kernel_Name']
    if 'pdwt' not in n: continue
    key=(n.split('(')[0].replace('void pdwt::',''), r['Grid_Size'])
    d[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items()):
    print(k, {c: '%.4g'%(sorted(x)[len(x)//2]) for c,x in v.items()})
PY
