#!/bin/bash
# One-box sweep of run-time switches on the C2 bench (2000 steps each, baseline repeated between variants).
# CFG=c3 STEPS=300 selects another config.
# usage: gpurun -- 'bash tools/env_sweep.sh "PDWT_CASC_WAVES=768" "PDWT_CASC_IWAVES=1024" ...'
export PYTHONPATH=$PWD
run() { env "$@" timeout 200 python bench.py --config ${CFG:-c2} --steps ${STEPS:-2000} --warmup 100 --cpu-seconds 0 --no-roofline 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us' % (d['ms_per_step']*1e3))"; }
echo -n "base: "; run A=1
for v in "$@"; do
  echo -n "$v: "; run $v
  echo -n "base: "; run A=1
done
