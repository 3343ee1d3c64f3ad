#!/bin/bash
# copy what tools/r6_final.sh left under gpurun_out/ into profiles/ (run in the build container after the gpurun call)
cd "$(dirname "$0")/.."
cp gpurun_out/profiles_r06/r06_*.md gpurun_out/profiles_r06/r06_*.json profiles/
cp gpurun_out/profiles_r06/pmc_traffic.json profiles/pmc_traffic.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
from bench import kernel_source_hash
for src, dst in (("bench_default", "r06_bench_default_run"), ("bench_k20", "r06_bench_driver_style_run")):
    d = json.loads(open('gpurun_out/r6f/%s.json' % src).read().strip().splitlines()[-1])
    json.dump(d, open('profiles/%s.json' % dst, 'w'), indent=1)
p = open('gpurun_out/r6f/pmc_lat.txt').read()
open('profiles/r06_c5_lat_pmc_sq.md', 'w').write("# r06: SQ counters of the level-1 lattice kernels (tools/r6_pmc_lat.sh lib: one rocprofv3 --pmc pass per group of four counters over `python tools/lat_time.py 1`; median per launch)\n\n```\n" + p.strip() + "\n```\n")
t = json.load(open('profiles/pmc_traffic.json')); h = kernel_source_hash()
print('source hash', h, 'all entries current:', all(e.get('src_sha16') == h for c in ('c2', 'c3', 'c4', 'c5') for e in t[c].values()))
PY
