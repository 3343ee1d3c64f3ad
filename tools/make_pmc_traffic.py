#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from two rocprofv3 PMC passes of `python bench.py` (same command,
separate passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2):
    tools/prof_pmc.sh f "FETCH_SIZE"          -> gpurun_out/pmc_f/t_counter_collection.csv
    tools/prof_pmc.sh w "WRITE_SIZE ..."      -> gpurun_out/pmc_w/t_counter_collection.csv
HBM bytes per launch of a kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over its launches in a
step.  The factor 2 is the gfx950 correction: FETCH_SIZE tallies 128-byte requests of wide coalesced
loads at 64 bytes (MI355X_MICROARCH.md, "HBM").  usage: tools/make_pmc_traffic.py <config> <fetch_dir> <write_dir>"""
import collections
import csv
import json
import os
import sys

cfg, fdir, wdir = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMAP = {"k_fwd2d_casc": "fwd2d_casc", "k_inv2d_casc": "inv2d_casc", "k_inv2d_cascw": "inv2d_casc", "k_inv2d_casc3": "inv2d_casc", "k_fwd2d_stream": "fwd2d_stream", "k_fwd2d_fused": "fwd2d_fused", "k_inv2d_stream": "inv2d_stream", "k_inv2d_fused": "inv2d_fused",
        "k_fwd2d_f64fused": "fwd2d_f64", "k_inv2d_f64fused": "inv2d_f64", "k_fwd2d_f64lds": "fwd2d_f64", "k_inv2d_f64lds": "inv2d_f64", "k_fwd2d_lat": "fwd2d_f64", "k_inv2d_lat": "inv2d_f64", "k_soft_thresh_sum": "thresh_sum", "k_soft_thresh": "soft_thresh", "k_abs_sum": "abs_sum",
        "k_ana_rows": "ana_rows", "k_ana_rows_tr": "ana_rows", "k_syn_rows_tr": "syn_rows", "k_ana_cols": "ana_cols", "k_syn_rows": "syn_rows", "k_syn_cols": "syn_cols",
        "k_fwd1d_stream": "ana_rows", "k_inv1d_stream": "syn_rows", "k_fwd1d_fused": "ana_rows", "k_inv1d_fused": "syn_rows", "k_inv1d_fused_pf": "syn_rows",
        "k_ana_cols_ring": "ana_cols", "k_ana_cols_ring_tr": "ana_cols", "k_syn_cols_ring": "syn_cols", "k_syn_cols_ring_tr": "syn_cols",
        "k_swt_fwd_fused": "swt_ana_cols", "k_swt_inv_fused": "swt_syn_cols", "k_swt_inv_fused4": "swt_syn_cols", "k_swt_inv_fusedp": "swt_syn_cols",
        "k_swt_ana_rows": "swt_ana_rows", "k_swt_ana_cols": "swt_ana_cols", "k_swt_syn_rows": "swt_syn_rows", "k_swt_syn_cols": "swt_syn_cols"}


def collect(d, counter):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(os.path.join(d, "t_counter_collection.csv"))):
        if r["Counter_Name"] != counter or "pdwt::" not in r["Kernel_Name"]:
            continue
        base = r["Kernel_Name"].split("<")[0].split("(")[0].split("::")[-1]  # (kernel name without namespaces: pdwt::nt::k_fwd1d_fused<...> -> k_fwd1d_fused)
        k = KMAP.get(base)
        if k:
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


fetch, nf = collect(fdir, "FETCH_SIZE")
write, _ = collect(wdir, "WRITE_SIZE")
out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
doc = json.load(open(out_path)) if os.path.exists(out_path) else {}
import subprocess
COMMIT = os.environ.get("PDWT_COMMIT", "")
try:  # (the GPU box has no .git: the caller passes the commit in PDWT_COMMIT)
    doc["commit"] = COMMIT or subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
except Exception:
    pass
doc["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 10 --warmup 3`; "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 averaged over the kernel's launches; summaries in profiles/*_pmc_*.md")
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402  (bench.py flags an entry whose hash differs from the sources it runs: traffic_stale)
SRC = kernel_source_hash()
doc[cfg] = {k: {"commit": COMMIT, "src_sha16": SRC, "hbm_bytes_per_launch": (2 * fetch[k] + write.get(k, 0.0)) * 1024, "fetch_kb_raw": fetch[k], "write_kb": write.get(k, 0.0), "launches_sampled": nf[k]}
            for k in fetch}
json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(doc[cfg], indent=1))
