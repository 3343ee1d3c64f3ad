#!/bin/bash
# A/B of variant builds (tools/build_variant.sh) against the default libraries on ONE box, interleaved:
#   gpurun -- 'CFG=c4 STEPS=200 REPS=3 bash tools/ab_variants.sh nt1 nt2 nt3'
export PYTHONPATH=$PWD
run() { env "$@" timeout 300 python bench.py --config ${CFG:-c4} --steps ${STEPS:-200} --warmup 20 --cpu-seconds 0 --no-roofline --no-others 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['ms_per_step'])"; }
for rep in $(seq 1 ${REPS:-3}); do
  echo -n "base: "; run A=1
  for v in "$@"; do echo -n "$v: "; run PDWT_LIBDIR=$PWD/pdwt_amd/lib_$v; done
done
