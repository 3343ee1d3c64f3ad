#!/bin/bash
# rocprofv3 kernel trace of an arbitrary command; prints per-(kernel,grid) median GPU durations
TAG=$1; shift
R=$PWD; mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$TAG -o t --output-format csv -- "$@" > $R/gpurun_out/prof_$TAG/cmd.log 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$R/gpurun_out/prof_$TAG/t_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    key=(n.split('(')[0][:40], r['Grid_Size_X'], r['Grid_Size_Y'])
    d[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items():
    v=sorted(v); print(k, 'n=%d med=%.1fus min=%.1f'%(len(v), v[len(v)//2]/1e3, v[0]/1e3))
PY
