#!/bin/bash
# the 20-step timed region of the driver's bench invocation under different wait policies of the HIP / HSA runtimes (one box, interleaved)
export PYTHONPATH=$PWD
run() { env "$@" python tools/k20_probe.py 2>/dev/null | awk '$1==20 && $2=="0:" {w+=$3; g+=$4; n++} $1==2000 {l=$3} END {printf "K=20 wall %.2f event %.2f (n=%d)   K=2000 wall %.2f\n", w/n, g/n, n, l}'; }
for rep in 1 2; do
  echo -n "default: "; run A=1
  for v in "$@"; do echo -n "$v: "; run $v; done
done
