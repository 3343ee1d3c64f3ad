#!/bin/bash
# GPU iteration helper for the headline path: cascade parity subset, then one bench line per environment variant.
# usage: tools/c2_iter.sh "VAR=1 OTHER=2" "VAR=3" ...   (each argument = one variant; "" = defaults)
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "${TESTS:-cascade or stress or config2 or determinism or golden_vectors}" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15
fi
[ $# -eq 0 ] && set -- ""
for V in "$@"; do
  echo "== [$V]"
  env $V python bench.py --config ${CONFIG:-c2} --steps ${STEPS:-1000} --warmup 100 --cpu-seconds 0 --no-others 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f  us/step %.2f (gpu %.2f)  dom %s %.0f GB/s frac %.3f  stepfrac %.3f  rt %.1e' % (d['value'], d['ms_per_step']*1e3, d['gpu_ms_per_step']*1e3, r['kernel'], r['achieved'], r['frac'], r['step_frac_of_peak'], d['roundtrip_max_rel_err']))
print('   ', {k:round(v['us_per_step'],2) for k,v in d['kernels'].items()})"
done
