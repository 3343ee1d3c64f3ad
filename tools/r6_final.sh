#!/bin/bash
# round-6 artefacts at the current tree: default bench line, driver-style line, profiles of every config, lattice-kernel PMC passes
mkdir -p gpurun_out/r6f
python bench.py > gpurun_out/r6f/bench_default.json 2> gpurun_out/r6f/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6f/bench_k20.json 2> gpurun_out/r6f/bench_k20.err
PDWT_COMMIT=${PDWT_COMMIT:-r06} timeout 1500 bash tools/make_all_profiles.sh r06 c2 c3 c4 c5 > gpurun_out/r6f/mkprof.log 2>&1
bash tools/r6_pmc_lat.sh lib > gpurun_out/r6f/pmc_lat.txt 2>&1
timeout 200 python tools/lds_trace.py --dir fwd > gpurun_out/r6f/lds_trace_fwd.txt 2>&1
timeout 200 python tools/lds_trace.py --dir inv > gpurun_out/r6f/lds_trace_inv.txt 2>&1
python - <<'PY'
import json
for f in ("bench_default","bench_k20"):
    d=json.loads(open('gpurun_out/r6f/%s.json'%f).read().strip().splitlines()[-1])
    print(f, 'C2', d['value'], d['ms_per_step'], 'steady', (d.get('steady_state') or {}).get('ms_per_step'), 'frac', d['roofline']['frac'], 'stepfrac', d['roofline']['step_frac_of_peak'], 'stream', d.get('value_streaming'), 'c4', d.get('c4_ms_per_step'))
    for k,v in (d.get('other_configs') or {}).items():
        r=v.get('roofline') or {}
        print('   ', k, v.get('ms_per_step'), v.get('value'), 'frac', r.get('frac'), 'stepfrac', r.get('step_frac_of_peak'), 'fp64', (r.get('fp64') or {}).get('frac'), (r.get('fp64') or {}).get('transform_kernels_us'), 'stale', r.get('traffic_stale'))
    print('    cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores'))
PY
