#!/bin/bash
# usage: tools/gpu_sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   -> one bench line per environment setting
for E in "$@"; do
  echo "== $E"; env $E python bench.py --steps ${STEPS:-200} --cpu-seconds 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f Mpix/s  ms/step %.4f  gpu_ms %.4f dom %s %.0f GB/s frac %.3f  stepfrac %.3f' % (d['value'], d['ms_per_step'], d['gpu_ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['step_frac_of_peak']), {k:round(v['us_per_step'],1) for k,v in d['kernels'].items()})"
done
