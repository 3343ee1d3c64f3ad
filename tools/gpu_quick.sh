#!/bin/bash
# quick GPU check used while iterating on kernels: core parity subset + bench line(s)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or fused_equals or dwt2_vs_oracle or config2 or state_machine" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15
for R in ${RLIST:-0}; do
  echo "== PDWT_STREAM_R=$R"; PDWT_STREAM_R=$R python bench.py --steps ${STEPS:-200} --cpu-seconds 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f Mpix/s  ms/step %.4f  dom %s %.0f GB/s frac %.3f  stepfrac %.3f  rt %.1e' % (d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['step_frac_of_peak'], d['roundtrip_max_rel_err']))
print({k:round(v['us_per_step'],1) for k,v in d['kernels'].items()})"
done
