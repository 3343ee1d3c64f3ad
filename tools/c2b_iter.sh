#!/bin/bash
# c2_batch (16 images, out of the Infinity Cache) in both forms: ImageBatch (one launch per kernel of the plan) and 16 cycled instances
for V in "" "--batch-instances"; do
  echo "== c2_batch [$V] $ENVV"
  env $ENVV python bench.py --config c2_batch --steps ${STEPS:-60} --warmup 10 --cpu-seconds 0 --no-others $V 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f  us/pair %.2f  dom %s %.0f GB/s frac %.3f  stepfrac %.3f  rt %.1e  copy %s' % (d['value'], d['ms_per_step']*1e3/16, r['kernel'], r['achieved'], r['frac'], r['step_frac_of_peak'], d['roundtrip_max_rel_err'], r.get('copy_ceiling')))
print('   ', {k:(round(v['us_per_step']/16,2), v['launches_per_step']) for k,v in d['kernels'].items()})"
done
