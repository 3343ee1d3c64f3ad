mkdir -p gpurun_out/r6a
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_batch_gpu.py -q -x -k "selfcheck_failed or plain_form or kernel_times_fit or handles_are_bound or config5" 2>&1 | tail -15 > gpurun_out/r6a/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r6a/bench_k20.json 2> gpurun_out/r6a/bench_k20.err
tail -5 gpurun_out/r6a/tests.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6a/bench_k20.json').read().strip().splitlines()[-1])
print('C2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('ktime_scale'))
for k,v in d['other_configs'].items():
    print(k, v.get('ms_per_step'), v.get('value'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('ktime_scale'), {a:b['us_per_step_timed'] for a,b in v.get('kernels',{}).items()})
PY
