#!/bin/bash
# C5 step under knob settings, interleaved on one box:  tools/r6_c5_sweep.sh "ENV=.. ENV=.." "..." (each argument one setting; "" = defaults)
for rep in 1 2 3; do
  for E in "$@"; do
    env $E python bench.py --config c5 --steps 30 --warmup 5 --no-others --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-44s ms/step %.4f  xf_us %6.1f  fp64 %.3f  %s' % ('$E' or 'defaults', d['ms_per_step'], r['fp64']['transform_kernels_us'], r['fp64']['frac'], {k:round(v['us_per_step_timed'],1) for k,v in d['kernels'].items() if 'f64' in k}))"
  done
done
