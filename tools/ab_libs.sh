#!/bin/bash
# A/B of two builds of the libraries on ONE box (launch durations drift ~25 % with the box and its power state, so
# numbers from different gpurun calls are not comparable).  Prepare `lib_old/` = a copy of pdwt_amd/lib built from the
# baseline sources (git stash; python -m pdwt_amd.build; cp -r pdwt_amd/lib lib_old; git stash pop; rebuild), then
#   gpurun -- 'bash tools/ab_libs.sh [bench args, default: --config c4]'
# alternates old/new three times and prints ms_per_step and the metric of each run.
ARGS=${@:---config c4}
export PYTHONPATH=$PWD
cp -r pdwt_amd/lib lib_new
for rep in 1 2 3; do
  for v in old new; do
    rm -rf pdwt_amd/lib; cp -r lib_$v pdwt_amd/lib
    echo -n "$v: "; timeout 200 python bench.py $ARGS --steps 200 --warmup 20 --cpu-seconds 0 --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
