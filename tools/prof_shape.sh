#!/bin/bash
# rocprofv3 kernel trace of tools/run_shape.py; per-(kernel,grid) median durations. Usage: tools/prof_shape.sh <tag> Nr Nc wname levels [reps] [dtype] [swt]
TAG=${1:-tmp}; shift
R=$PWD; mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o t --output-format csv -- python $R/tools/run_shape.py "$@" > $R/gpurun_out/prof_$TAG/run.log 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$R/gpurun_out/prof_$TAG/t_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'pdwt' not in n: continue
    key=(n.split('(')[0].replace('void pdwt::',''), r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'], 'v'+r['VGPR_Count'], 'lds'+r['LDS_Block_Size'])
    d[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=0
for k,v in sorted(d.items()):
    v=sorted(v); tot+=v[len(v)//2]; print(k, 'n=%d med=%.1fus min=%.1f'%(len(v), v[len(v)//2]/1e3, v[0]/1e3))
print('sum of medians %.1f us'%(tot/1e3))
PY
