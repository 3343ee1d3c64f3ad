#!/bin/bash
# Sample clocks / power with rocm-smi while a bench config runs (is a kernel's sustained rate set by the power state?).
# usage: gpurun -- 'bash tools/clock_watch.sh c4 20000'
CFG=${1:-c2}; STEPS=${2:-20000}
export PYTHONPATH=$PWD
python bench.py --config $CFG --steps $STEPS --warmup 10 --cpu-seconds 0 --no-roofline > /tmp/cw.log 2>&1 &
BP=$!
sleep 6   # torch import + settle
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g; s/ \+/ /g'; echo
  sleep 0.5
done
wait $BP
grep '^{' /tmp/cw.log | tail -1 | cut -c1-220
