#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PDWT hot path on MI355X.

Metric (BASELINE.json): Mpixels/s of one forward()+inverse() pair of the separable DWT,
4096x4096 float32, db4, 3 levels (configs[1]), inputs resident in HBM when the timed region starts,
plus the achieved HBM GB/s of the dominant kernel against the chip's 8 TB/s peak.

A "step" = one Wavelets.forward() + Wavelets.inverse() on one image per GPU.  N > 1 (launched by
torch.distributed.run, one rank per GPU) is the batch split of BASELINE.json: whole images are
independent, so every rank transforms its own image -- no data-path collective -- and the job
throughput is N images per step ("scaling": "weak").  RCCL is used only for the barriers and the
max-over-ranks of the elapsed time.

Extra objects in the JSON line:
  roofline      dominant kernel: algorithmic bytes of its launches / their HIP-event duration
                (events recorded on the library stream by pdwt_ktime_*), vs 8000 GB/s.
  cpu_baseline  the CPU oracle (plain-C restatement, OpenMP) timed on this host's cores on a
                bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured float4 copy
FP64_VECTOR_PEAK_TFLOPS = 78.6  # half the 157.3 TFLOP/s FP32 vector peak of MI355X_MICROARCH.md (ridge 9.8 flop/B)

# name -> (rows per GPU, cols, dtype, wavelet, levels, do_swt, ndim, extra ops, unit)
CONFIGS = {
    "c2": dict(Nr=4096, Nc=4096, dtype="float32", wname="db4", levels=3, do_swt=0, ndim=2, extra=False, unit="Mpixels/s",
               desc="4096x4096 float32, db4, 3 levels, separable DWT fwd+inv (BASELINE.json configs[1])"),
    # the headline workload on a BATCH of distinct images cycled per step: the single-image working set (~170 MB) lives in the 256 MiB
    # Infinity Cache, a batch that rotates through > 1 GB is what streams through HBM (report both)
    "c2_batch": dict(Nr=4096, Nc=4096, dtype="float32", wname="db4", levels=3, do_swt=0, ndim=2, extra=False, unit="Mpixels/s", batch=16,
                     desc="16 distinct 4096x4096 float32 images per GPU cycled per step, db4, 3 levels, fwd+inv each (working set 16 x 268 MB: out of the Infinity Cache)"),
    "c3": dict(Nr=4096, Nc=4096, dtype="float32", wname="db7", levels=5, do_swt=1, ndim=2, extra=False, unit="Mpixels/s",
               desc="4096x4096 float32, db7, 5 levels, SWT fwd+inv (configs[2])"),
    "c4": dict(Nr=8192, Nc=8192, dtype="float32", wname="sym8", levels=4, do_swt=0, ndim=1, extra=False, unit="Msamples/s",
               desc="batched-1D shard: 8192 signals x 8192 samples per GPU (65536 over 8 GPUs), sym8, 4 levels (configs[3])"),
    "c5": dict(Nr=8192, Nc=8192, dtype="float64", wname="db20", levels=6, do_swt=0, ndim=2, extra=True, unit="Mpixels/s",
               desc="8192x8192 float64, db20, 6 levels + soft_threshold + norm1 (configs[4])"),
}


def algorithmic_bytes(cfg, levels_eff):
    """Per-step ALGORITHMIC (compulsory) bytes, SURVEY.md 8(d), and per-kernel-launch-set figures.

    whole step : DWT fwd+inv = 4*N*sizeof(T); SWT-2D = 2*(3L+2)*N*sizeof(T); SWT-1D = 2*(L+2)*N*sizeof
    per kernel : each level kernel must read its inputs once and write its outputs once:
                 fused 2D DWT level on an n-element input: (n + 4*n/4)*sizeof = 2*n*sizeof, both directions;
                 1-D level: 2*n*sizeof; SWT pass kernels: rows (1 in, 2 out), cols (2 in, 4 out), ...
    """
    import numpy as np
    sz = np.dtype(cfg["dtype"]).itemsize
    N = cfg["Nr"] * cfg["Nc"]
    L = levels_eff
    per_kernel = {}
    if cfg["do_swt"]:
        if cfg["ndim"] == 2:
            step = 2 * (3 * L + 2) * N * sz
            per_kernel = {"swt_ana_rows": 3 * N * sz * L, "swt_ana_cols": 6 * N * sz * L, "swt_syn_cols": 6 * N * sz * L, "swt_syn_rows": 3 * N * sz * L}
        else:
            step = 2 * (L + 2) * N * sz
            per_kernel = {"swt_ana_rows": 3 * N * sz * L, "swt_syn_rows": 3 * N * sz * L}
    else:
        step = 4 * N * sz
        n = 0
        r, c = cfg["Nr"], cfg["Nc"]
        for _ in range(L):
            n += r * c
            if cfg["ndim"] == 2:
                r = (r + 1) // 2
            c = (c + 1) // 2
        if cfg["ndim"] == 2:
            per_kernel = {"fwd2d_fused": 2 * n * sz, "inv2d_fused": 2 * n * sz,
                          "ana_rows": 2 * n * sz, "ana_cols": 2 * n * sz, "syn_cols": 2 * n * sz, "syn_rows": 2 * n * sz,
                          "haar2d_fwd": 2 * n * sz, "haar2d_inv": 2 * n * sz}
            # float32 path: levels (1,2) run as ONE cascade launch (dwt_casc.hip) that reads the N-element input once and
            # writes 3N/4 + 4N/16 = N coefficients (the level-1 approximation never goes to memory); the remaining
            # levels are one launch each
            if sz == 4 and L >= 2:
                n12 = N + ((cfg["Nr"] + 1) // 2) * ((cfg["Nc"] + 1) // 2)
                per_kernel.update({"fwd2d_casc": 2 * N * sz, "inv2d_casc": 2 * N * sz,
                                   "fwd2d_fused|casc": 2 * (n - n12) * sz, "inv2d_fused|casc": 2 * (n - n12) * sz})
        else:
            # batched 1D runs ALL levels in one launch (dwt1d_fused.hip): read the batch once, write every band once
            per_kernel = {"ana_rows": 2 * N * sz, "syn_rows": 2 * N * sz, "haar1d_fwd": 2 * n * sz, "haar1d_inv": 2 * n * sz}
    return step, per_kernel


def usable_cores():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota of the container (the round-4 GPU boxes show
    256 hardware threads but grant `cpu.max = 1600000 100000` = 16 CPUs: every OpenMP team beyond that is throttled, which is what the
    collapse of round 4's `by_threads` above 16 threads was)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return (max(1, min(n, int(quota + 0.5))) if quota else n), n, quota


_CPU_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
cfg = json.loads(sys.argv[2]); Nr, Nc, cands, per = int(sys.argv[3]), int(sys.argv[4]), json.loads(sys.argv[5]), float(sys.argv[6])
import numpy as np
from oracle import oracle as orc
x = np.random.RandomState(0).uniform(0, 255, (Nr, Nc)).astype(cfg["dtype"])
W = orc.OracleWavelets(x, cfg["wname"], cfg["levels"], do_swt=cfg["do_swt"], ndim=cfg["ndim"])
def pair():
    W.forward()
    if cfg["extra"]:
        W.soft_threshold(0.5)
        W.norm1_f64()
    W.inverse()
res = {}
for nthreads in cands:
    used = orc.set_num_threads(nthreads)
    pair()  # warm (page faults -- first touch by the team that will use the pages --, thread team)
    t0 = time.perf_counter(); pair(); t1 = time.perf_counter() - t0
    reps = int(max(1, min(200, per / max(t1, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        pair()
    dt = (time.perf_counter() - t0) / reps
    res[used] = dict(value=Nr * Nc / dt / 1e6, threads=used, reps=reps, s_per_pair=dt)
print(json.dumps({"res": res, "levels": W.info.nlevels}))
"""


def cpu_baseline(cfg, seconds_budget):
    """Time the CPU oracle (kind 'port') on this host's cores on a bounded sample of the workload.  Runs in a process of its own: the
    bench process has torch's libgomp loaded and configured, and OpenMP reads its environment once -- the child gets passive waiting
    (a team larger than the CPU quota must not spin its quota away) and spread binding over the cores."""
    import subprocess
    Nr, Nc = cfg["Nr"], cfg["Nc"]
    scale = 1
    # keep one pair at a few seconds at most: shrink the sample for the heavy configs
    while cfg["do_swt"] and Nr * Nc > 1024 * 1024:
        Nr //= 2
        Nc //= 2
        scale *= 4
    if cfg["dtype"] == "float64" and Nr * Nc > 4096 * 4096:
        Nr //= 2
        Nc //= 2
        scale *= 4
    usable, affinity, quota = usable_cores()
    host = os.cpu_count() or 1
    # single thread, the CPUs the container is granted ("all cores" of THIS process), and teams below / beyond that for the shape of the curve
    cands = sorted({1, usable} | {t for t in (usable // 2, 2 * usable, 4 * usable) if 1 <= t <= host})
    per = seconds_budget / (len(cands) + 2.0)
    env = dict(os.environ, OMP_WAIT_POLICY="passive", OMP_PROC_BIND="spread", OMP_PLACES="cores")
    env.pop("OMP_NUM_THREADS", None)
    ccfg = {k: cfg[k] for k in ("dtype", "wname", "levels", "do_swt", "ndim", "extra")}
    out = subprocess.run([sys.executable, "-c", _CPU_CHILD, ROOT, json.dumps(ccfg), str(Nr), str(Nc), json.dumps(cands), str(per)],
                         capture_output=True, text=True, timeout=max(120.0, 20.0 * seconds_budget), env=env)
    if out.returncode != 0:
        raise RuntimeError("cpu_baseline child failed: " + out.stderr[-500:])
    j = json.loads(out.stdout.strip().splitlines()[-1])
    res = {int(k): v for k, v in j["res"].items()}
    best = max(res.values(), key=lambda r: r["value"])
    pywt_ref = pywt_timing(cfg, Nr, Nc) if not cfg["extra"] else None
    return {
        "pywt": pywt_ref,
        "value": round(best["value"], 2), "unit": cfg["unit"], "cores": best["threads"], "kind": "port",
        "sample": "%d x fwd+inv of a %dx%d %s %s L%d input (%s of the GPU workload's pixels per pair), oracle/pdwt_oracle.c with OpenMP "
                  "(own process: OMP_WAIT_POLICY=passive, OMP_PROC_BIND=spread, OMP_PLACES=cores), best of thread counts %s"
                  % (best["reps"], Nr, Nc, cfg["dtype"], cfg["wname"], j["levels"], "all" if scale == 1 else "1/%d" % scale, sorted(res)),
        "single_thread_value": round(res[1]["value"], 2),
        "all_cores_value": round(res[usable]["value"], 2) if usable in res else None, "usable_cores": usable,
        "host_cores": host, "affinity_cores": affinity, "cgroup_cpu_quota": quota,
        "note": "`value` is the BEST OF the thread counts in by_threads, not an all-host-cores figure: usable_cores = min(affinity, cgroup cpu.max quota) "
                "is what 'all cores' means for this process (the GPU boxes grant the container a quota of %s CPUs of a %d-thread host; teams beyond "
                "the quota share it, and the host's other cores were never timed)" % ("%g" % quota if quota else "all", host),
        "by_threads": {str(k): round(v["value"], 1) for k, v in sorted(res.items())},
    }


def pywt_timing(cfg, Nr, Nc):
    """PyWavelets (the library BASELINE.json names as the parity reference) timed on the same sample, when a
    python with pywt exists on this host ("probe at run time, never depend on it").  Single-threaded
    by construction; None when unavailable.  Not the oracle and not the cpu_baseline value: context only."""
    import subprocess
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or cfg["ndim"] != 2:
        return None
    code = ("import time,numpy as np,pywt\n"
            "x=np.random.RandomState(0).uniform(0,255,(%d,%d)).astype('%s')\n"
            "f=(lambda: pywt.iswt2(pywt.swt2(x,'%s',%d),'%s')) if %d else (lambda: pywt.waverec2(pywt.wavedec2(x,'%s','periodization',%d),'%s','periodization'))\n"
            "f();t=time.perf_counter();f();print(time.perf_counter()-t)\n"
            % (Nr, Nc, cfg["dtype"], cfg["wname"], cfg["levels"], cfg["wname"], cfg["do_swt"], cfg["wname"], cfg["levels"], cfg["wname"]))
    try:
        out = subprocess.run([py, "-c", code], capture_output=True, text=True, timeout=90)
        dt = float(out.stdout.strip().splitlines()[-1])
        return {"value": round(Nr * Nc / dt / 1e6, 2), "unit": cfg["unit"], "cores": 1, "s_per_pair": round(dt, 4),
                "what": "pywt wavedec2+waverec2 (or swt2+iswt2) on the same %dx%d sample" % (Nr, Nc)}
    except Exception:
        return None


def kernel_source_hash():
    """sha256 (16 hex digits) over the kernel sources of the library: profiles/pmc_traffic.json keeps the value its counters were
    collected at, so a static traffic figure that predates a kernel change is flagged (`traffic_stale`) instead of trusted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pdwt_amd", "csrc", "*"))):
        if f.endswith((".hip", ".inc", ".hpp")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pick_dominant(cfg, kernels, per_kernel_bytes, levels_eff):
    """Dominant kernel (largest time per step among those with an algorithmic byte count) -> (name, bytes per step)."""
    import numpy as np
    if cfg["do_swt"] and cfg["ndim"] == 2 and "swt_ana_rows" not in kernels and "swt_ana_cols" in kernels:
        # float32 SWT levels run as ONE launch per direction (swt_fused.inc, timed under the *_cols ids): a level reads N
        # samples and writes 4N (forward) / reads 4N and writes N (inverse)
        nb = 5 * cfg["Nr"] * cfg["Nc"] * np.dtype(cfg["dtype"]).itemsize * levels_eff
        per_kernel_bytes["swt_ana_cols"] = per_kernel_bytes["swt_syn_cols"] = nb
    for d in ("fwd2d", "inv2d"):
        if d + "_casc" in kernels and d + "_fused|casc" in per_kernel_bytes:
            # single-level launches cover only the levels the cascade launches did not; whichever single-level kernels ran
            # (streaming / register-tile / LDS-tiled) share those levels -- the byte count goes to the one that took longest
            rest = [k for k in (d + "_stream", d + "_small", d + "_fused") if k in kernels]
            if rest:
                per_kernel_bytes[max(rest, key=lambda k: kernels[k]["us_per_step"])] = per_kernel_bytes[d + "_fused|casc"]
        elif d + "_fused" in per_kernel_bytes:
            for k in (d + "_stream", d + "_f64"):
                if k in kernels and d + "_fused" not in kernels:
                    per_kernel_bytes[k] = per_kernel_bytes[d + "_fused"]
    cand = [k for k in kernels if k in per_kernel_bytes]
    if not cand:
        return None, 0
    dom = max(cand, key=lambda k: kernels[k]["us_per_step"])
    return dom, per_kernel_bytes[dom]


def run_config(name, args, L, torch, dist, rank, world, steps, warmup, settle_ms, cpu_seconds, do_roofline):
    """Time `steps` steps of config `name` on this rank's GPU; returns the fields of the JSON line for it."""
    import numpy as np
    cfg = CONFIGS[name]

    def barrier():
        if world > 1:
            dist.barrier()

    # synthetic input generated on the GPU that owns it (uniform [0,255), seed = rank)
    tdt = torch.float32 if cfg["dtype"] == "float32" else torch.float64
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    img = torch.rand(cfg["Nr"], cfg["Nc"], device="cuda", dtype=tdt, generator=g) * 255.0
    torch.cuda.synchronize()
    B = None
    if world > 1:
        # N > 1: the batch split of the product (pdwt_amd/batch.py).  The global batch is `world` x this config's per-GPU block
        # (whole images, or rows of the batched-1D array: shard_rows(world * Nr, world, rank) = Nr rows each); every rank
        # transforms its own shard with a private Wavelets on its own GPU, no data-path collective.
        from pdwt_amd.batch import ShardedBatch, shard_rows
        assert shard_rows(world * cfg["Nr"], world, rank) == (rank * cfg["Nr"], cfg["Nr"])
        B = ShardedBatch(img, cfg["wname"], cfg["levels"], ndim=cfg["ndim"], do_swt=cfg["do_swt"])
        W = B.W
    else:
        W = pdwt_amd_mod().Wavelets(None, cfg["wname"], cfg["levels"], do_swt=cfg["do_swt"], ndim=cfg["ndim"], dtype=cfg["dtype"],
                                    shape=(cfg["Nr"], cfg["Nc"]), device_ptr=img.data_ptr())
    assert W.state == pdwt_amd_mod().W_INIT, "Wavelets creation failed"
    if cfg["extra"]:
        W.set_norm_cache(False)  # the drop-in class's default (reference behaviour): norm1() reduces the bands on every call
    levels_eff = W.info.nlevels
    nbatch = cfg.get("batch", 1)
    Ws = [W]
    IB = None
    if nbatch > 1 and not getattr(args, "batch_instances", False):
        # the product's batched entry (include/wt_batch.h WaveletsImages / pdwt_batch2d_*): all images in ONE launch per kernel of the
        # level plan (gridDim.y = image) -- at this size the two cascade launches + the level-3 launch of the single image
        stack = torch.empty(nbatch, cfg["Nr"], cfg["Nc"], device="cuda", dtype=tdt)
        for bi in range(nbatch):
            g.manual_seed(1234 + rank + 1000 * bi)
            stack[bi] = torch.rand(cfg["Nr"], cfg["Nc"], device="cuda", dtype=tdt, generator=g) * 255.0
        torch.cuda.synchronize()
        IB = pdwt_amd_mod().ImageBatch(stack, cfg["wname"], cfg["levels"])
        del stack
    for bi in range(1, nbatch if IB is None else 1):  # the other images of the batch: distinct data, private instances (image + bands + scratch each)
        g.manual_seed(1234 + rank + 1000 * bi)
        xi = torch.rand(cfg["Nr"], cfg["Nc"], device="cuda", dtype=tdt, generator=g) * 255.0
        torch.cuda.synchronize()
        Ws.append(pdwt_amd_mod().Wavelets(None, cfg["wname"], cfg["levels"], do_swt=cfg["do_swt"], ndim=cfg["ndim"], dtype=cfg["dtype"],
                                          shape=(cfg["Nr"], cfg["Nc"]), device_ptr=xi.data_ptr()))
        del xi

    if IB is not None:
        def step():
            IB.forward()
            IB.inverse()
    elif nbatch > 1:
        def step():
            for Wi in Ws:
                Wi.forward()
                Wi.inverse()
    elif cfg["extra"]:
        def step():
            W.forward()
            W.soft_threshold(0.5)
            W.norm1()
            W.inverse()
    else:
        def step():
            W.forward()
            W.inverse()

    def sync():
        L.pdwt_sync()
        torch.cuda.synchronize()

    # untimed: bring the box to its steady clocks first (it needs ~50-100 ms of load: 20 / 2000 timed steps of C2 gave
    # 62.4 / 59.7 us per step on one box without this), then the W warm-up steps of the contract
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
        for _ in range(20):
            step()
        sync()
    for _ in range(warmup):
        step()
    sync()
    barrier()
    sync()
    # the timed region of the contract: barrier + synchronize on both sides and NOTHING but the K steps inside -- no event markers
    # (two barrier packets on the queue: +0.16 us per step at K = 20) and one device-wide synchronize, which covers the library's
    # stream (a stream synchronize in front of it: +0.3 us per step at K = 20; tools/k20_bracket.py)
    wall0 = time.time()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    wall1 = time.time()
    power = _SAMPLER.window(wall0, wall1) if _SAMPLER is not None else None
    # the same K steps once more between two HIP events on the library stream: `gpu_ms_per_step` (what the GPU spent between the first
    # kernel's start and the last one's end, without the bracket)
    sync()
    e0, e1 = L.pdwt_event_create(), L.pdwt_event_create()
    L.pdwt_event_record(e0)
    for _ in range(steps):
        step()
    L.pdwt_event_record(e1)
    sync()
    gpu_ms = L.pdwt_event_elapsed_ms(e0, e1)
    # A short timed region (the driver runs `--steps 20 --warmup 5`: 1 ms of C2) carries the bracket itself -- first launch on an idle GPU,
    # wake-up of the final synchronize: ~25 us, 1.2 us per step at K = 20 (tools/k20_probe.py) -- and a cold start of the caches.  The
    # contract's number stays `ms_per_step`; what the same loop costs per step once that is amortised is reported NEXT to it (untimed for
    # `value`): `steady_state` = the same step repeated until >= 0.1 s of GPU work, bracketed the same way.
    steady = None
    if world == 1 and elapsed < 0.05:
        n_long = max(200, min(5000, int(0.1 / max(elapsed / steps, 1e-6))))
        sync()
        t0s = time.perf_counter()
        for _ in range(n_long):
            step()
        torch.cuda.synchronize()
        steady = {"steps": n_long, "ms_per_step": round((time.perf_counter() - t0s) / n_long * 1e3, 5),
                  "what": "the same step, same bracket, over a region long enough to amortise the bracket (first launch on an idle GPU + synchronize wake-up) and the cold start; not used for `value`"}
    if world > 1:
        t = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity of the timed work (untimed): pixels really go through.  Every config: one fresh forward+inverse reproduces the
    # input.  Configs with the extra operators: norm1() against a float64 sum over the bands taken by torch on the GPU
    # (zero-copy views), before and after the threshold, and the thresholded norm must be the smaller one.
    sanity = {}
    ref = img.cpu().numpy()
    # (a separate instance: the zero-copy band views below hand out raw pointers, which switches off an instance's
    # threshold-time norm bookkeeping -- the timed instance must stay as it was for the per-kernel pass that follows)
    S = pdwt_amd_mod().Wavelets(None, cfg["wname"], cfg["levels"], do_swt=cfg["do_swt"], ndim=cfg["ndim"], dtype=cfg["dtype"],
                                shape=(cfg["Nr"], cfg["Nc"]), device_ptr=img.data_ptr())
    S.forward()
    if cfg["extra"]:
        def torch_norm1(X):
            X.sync()
            return float(sum(torch.as_tensor(X.coeff_view(k), device="cuda").abs().sum(dtype=torch.float64).item() for k in range(X.nbands)))
        S2 = S.copy()  # thresholded through the same call sequence as the timed step, on an instance nobody took pointers of
        S2.soft_threshold(0.5)
        n_lib2 = S2.norm1_f64()
        n_ref2 = torch_norm1(S2)
        n_lib, n_ref = S.norm1_f64(), torch_norm1(S)
        sanity["norm1_rel_err"] = abs(n_lib - n_ref) / n_ref
        sanity["norm1_after_threshold_rel_err"] = abs(n_lib2 - n_ref2) / n_ref2
        sanity["threshold_shrinks_norm1"] = bool(n_lib2 < n_lib)
        del S2
    S.inverse()
    out = S.get_image()
    del S
    rt_err = float(np.abs(out.astype(np.float64) - ref).max() / np.abs(ref).max())
    sanity["roundtrip_max_rel_err"] = rt_err
    sanity["roundtrip_mean_rel_err"] = float(np.abs(out.astype(np.float64) - ref).mean() / np.abs(ref).max())
    if B is not None:
        # untimed check of the one exchange step of the N > 1 path: norm1() = all-reduce(SUM) of one double per rank (RCCL)
        W.forward()
        n_all, parts = B.norm1(), B.norm1_per_rank()
        sanity["norm1_allreduce_rel_err"] = abs(sum(parts) - n_all) / n_all
        sanity["norm1_ranks"] = len(parts)
        W.inverse()

    pixels = cfg["Nr"] * cfg["Nc"] * nbatch
    ms_per_step = elapsed / steps * 1e3
    value = world * pixels / (elapsed / steps) / 1e6
    step_bytes, per_kernel_bytes = algorithmic_bytes(cfg, levels_eff)
    step_bytes *= nbatch
    per_kernel_bytes = {k: v * nbatch for k, v in per_kernel_bytes.items()}

    roofline = None
    kernels = {}
    if do_roofline:
        L.pdwt_ktime_enable(1)
        L.pdwt_ktime_reset()
        ksteps = min(steps, 200)
        ke0, ke1 = L.pdwt_event_create(), L.pdwt_event_create()
        L.pdwt_event_record(ke0)
        for _ in range(ksteps):
            step()
        L.pdwt_event_record(ke1)
        sync()
        # One clock for the roofline fields.  The per-kernel pass carries two events per dispatch and may run at another clock than the
        # timed pass (C4's kernel sum exceeded ms_per_step).  Its own duration is reported (`ktime_pass_ms_per_step`), and every
        # figure derived from a kernel time (`roofline.achieved / frac`, `us_per_step_timed`) is brought onto the clock of the timed step:
        # when the kernels' sum exceeds `gpu_ms_per_step` they are scaled down together so that sum(us_per_step_timed) <= gpu_ms_per_step.
        ktime_pass_ms = L.pdwt_event_elapsed_ms(ke0, ke1) / ksteps
        n, ms = C.c_int(), C.c_double()
        ksum_ms = 0.0
        for k in range(L.pdwt_kernel_count()):
            L.pdwt_ktime_read(k, C.byref(n), C.byref(ms))
            ksum_ms += ms.value / ksteps if n.value else 0.0
        # (the pass also holds the event packets between the kernels, so kernel times are only ever scaled DOWN, and only when their sum
        #  does not fit the timed step: the per-kernel pass ran slower than the timed one)
        kscale = min(1.0, (gpu_ms / steps) / ksum_ms) if ksum_ms > 0 else 1.0
        for k in range(L.pdwt_kernel_count()):
            L.pdwt_ktime_read(k, C.byref(n), C.byref(ms))
            if n.value:
                kernels[L.pdwt_kernel_name(k).decode()] = dict(launches_per_step=n.value / ksteps, us_per_step=ms.value * 1e3 / ksteps,
                                                               avg_us=ms.value * 1e3 / n.value, us_per_step_timed=ms.value * 1e3 / ksteps * kscale)
        L.pdwt_ktime_enable(0)
        L.pdwt_ktime_reset()
        dom, kb = pick_dominant(cfg, kernels, per_kernel_bytes, levels_eff)
        if dom:
            ach = kb / (kernels[dom]["us_per_step_timed"] * 1e-6) / 1e9
            # HBM bytes per launch: NOT measured in this run (PMC counters need rocprofv3 passes of their own).  The figure is
            # the one committed under profiles/ for this kernel -- (2*FETCH_SIZE + WRITE_SIZE) per the gfx950 correction of
            # MI355X_MICROARCH.md -- marked static with the commit it was taken at; null when the kernel has none.
            traffic, tsrc, tcommit, tstale = None, None, None, None
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    ent = tj.get("c2" if name == "c2_batch" else name, {}).get(dom)
                    if ent:
                        traffic, tsrc, tcommit = ent["hbm_bytes_per_launch"], tj.get("source"), ent.get("commit", tj.get("commit"))
                        # stale = the kernel sources changed since the counters were collected (hash kept with the entry)
                        tstale = ent.get("src_sha16") != kernel_source_hash()
                except Exception:
                    pass
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_static": traffic is not None,
                        "traffic_commit": tcommit, "traffic_stale": tstale, "traffic_source": tsrc,
                        "algorithmic_bytes_per_launch": kb / kernels[dom]["launches_per_step"],
                        "avg_launch_us": round(kernels[dom]["avg_us"] * kscale, 2), "avg_launch_us_ktime_pass": round(kernels[dom]["avg_us"], 2),
                        "launches_per_step": kernels[dom]["launches_per_step"],
                        "ktime_pass_ms_per_step": round(ktime_pass_ms, 5), "ktime_scale": round(kscale, 4),
                        "clock": "kernel times come from the per-kernel pass (events on every dispatch: ktime_pass_ms_per_step); when their sum exceeds the timed step (gpu_ms_per_step) they are scaled down together (ktime_scale < 1)",
                        "step_compulsory_bytes": step_bytes,
                        "step_compulsory_GBps": round(step_bytes / (gpu_ms / steps * 1e-3) / 1e9, 1),
                        "step_frac_of_peak": round(step_bytes / (gpu_ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}

    if roofline is not None and rank == 0:
        # What a plain device-to-device copy of the SAME byte volume reaches on this box, measured now (torch's copy kernel, half the
        # bytes read + half written): the spec peak is not reachable by any stream here, and the reachable rate depends on whether
        # the working set fits the 256 MB Infinity Cache (tools/copy_ceiling.py: 6.6-7.2 TB/s up to 256 MB moved, 4.7-4.9 TB/s beyond 1 GB).
        try:
            nbytes = int(roofline["algorithmic_bytes_per_launch"]) // 2
            # (a batch config rotates through as many distinct buffer pairs as it has images, so that the copy streams through HBM too)
            pairs = []
            for _ in range(nbatch):
                a_ = torch.empty(nbytes // 4, device="cuda", dtype=torch.float32).normal_()
                pairs.append((a_, torch.empty_like(a_)))
            for a_, b_ in pairs[:3]:
                b_.copy_(a_)
            torch.cuda.synchronize()
            ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            creps = max(30 // nbatch, 2)
            ce0.record()
            for _ in range(creps):
                for a_, b_ in pairs:
                    b_.copy_(a_)
            ce1.record()
            torch.cuda.synchronize()
            tgbps = 2.0 * nbytes * nbatch / (ce0.elapsed_time(ce1) * 1e-3 / creps) / 1e9
            # the same volume through the library's own copy kernel (pdwt_probe_bandwidth: 16-byte accesses, one contiguous chunk per
            # workgroup -- the walk that streams best on this chip, profiles/r04_hbm_ceiling.md), plus its read-only and write-only halves
            def probe(mode):
                pe0, pe1 = L.pdwt_event_create(), L.pdwt_event_create()
                for a_, b_ in pairs[:2]:
                    L.pdwt_probe_bandwidth(C.c_void_p(a_.data_ptr()), C.c_void_p(b_.data_ptr()), nbytes, mode)
                L.pdwt_sync()
                L.pdwt_event_record(pe0)
                for _ in range(creps):
                    for a_, b_ in pairs:
                        L.pdwt_probe_bandwidth(C.c_void_p(a_.data_ptr()), C.c_void_p(b_.data_ptr()), nbytes, mode)
                L.pdwt_event_record(pe1)
                L.pdwt_sync()
                return (2.0 if mode == 0 else 1.0) * nbytes * nbatch / (L.pdwt_event_elapsed_ms(pe0, pe1) * 1e-3 / creps) / 1e9
            pgbps, rgbps, wgbps = probe(0), probe(1), probe(2)
            cgbps = max(tgbps, pgbps)
            roofline["copy_ceiling"] = {"GBps": round(cgbps, 1), "first_party_copy_GBps": round(pgbps, 1), "torch_copy_GBps": round(tgbps, 1),
                                        "first_party_read_only_GBps": round(rgbps, 1), "first_party_write_only_GBps": round(wgbps, 1),
                                        "bytes_moved": 2 * nbytes, "buffer_pairs_cycled": nbatch,
                                        "achieved_over_copy": round(roofline["achieved"] / cgbps, 4),
                                        "what": "device-to-device copy of the dominant kernel's algorithmic byte volume (per image), measured in this run: "
                                                "the faster of the library's own copy kernel (pdwt_probe_bandwidth) and torch's"}
            del pairs
        except Exception as e:
            roofline["copy_ceiling"] = {"error": repr(e)}

    if roofline is not None and cfg["dtype"] == "float64" and not cfg["do_swt"]:
        # long double-precision banks are FP64-VALU-bound on compulsory bytes -> also report the FP64 fraction.
        # forward FMAs per level = hlen*(2*r*c2 + 4*r2*c2) (2-D) or hlen*2*r*c2 (1-D); fwd+inv = 4x that in flop
        hl, r, c, fma = W.info.hlen, cfg["Nr"], cfg["Nc"], 0
        for _ in range(levels_eff):
            c2 = (c + 1) // 2
            if cfg["ndim"] == 2:
                r2 = (r + 1) // 2
                fma += hl * (2 * r * c2 + 4 * r2 * c2)
                r = r2
            else:
                fma += hl * 2 * r * c2
            c = c2
        # transform kernels only (measured kernel time, threshold and norm excluded) and the whole step
        t_xf = sum(v["us_per_step_timed"] for k, v in kernels.items() if k not in ("soft_thresh", "abs_sum", "abs_sum_final", "thresh_sum")) * 1e-6
        tf_step = 4.0 * fma / (gpu_ms / steps * 1e-3) / 1e12
        tf_xf = 4.0 * fma / t_xf / 1e12 if t_xf > 0 else None
        roofline["fp64"] = {"flop_per_step": 4 * fma, "achieved": round(tf_xf, 2) if tf_xf else None, "peak": FP64_VECTOR_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(tf_xf / FP64_VECTOR_PEAK_TFLOPS, 4) if tf_xf else None,
                            "transform_kernels_us": round(t_xf * 1e6, 1),
                            "whole_step_achieved": round(tf_step, 2), "whole_step_frac": round(tf_step / FP64_VECTOR_PEAK_TFLOPS, 4),
                            "note": "achieved/frac: flops over the measured time of the transform kernels; whole_step_*: over the step (threshold and norm included)"}
        roofline["hbm_transform_kernels"] = {"compulsory_bytes": 4 * pixels * 8, "GBps": round(4 * pixels * 8 / t_xf / 1e9, 1) if t_xf > 0 else None,
                                             "frac": round(4 * pixels * 8 / t_xf / 1e9 / HBM_PEAK_GBPS, 4) if t_xf > 0 else None}

    clock_probe = None
    if cfg["dtype"] == "float64" and not cfg["do_swt"] and cfg["ndim"] == 2 and hasattr(L, "pdwt_clock_probe_enable"):
        # In-kernel clock of the level-1 launches (the probe of dwt_lds.hip: shader-clock counter over the 100 MHz counter, workgroup 0)
        # next to what amdsmi reports for the SAME window -- VERDICT r2: the two instruments disagreed (1.4-1.5 GHz vs 2.29 GHz)
        L.pdwt_clock_probe_enable(1)
        for _ in range(3):
            step()
        sync()
        SLOTS = (("fwd_level1", 1), ("inv_level1", 9), ("fwd_level2", 2), ("inv_level2", 10))

        def read_slots():
            out = {}
            for nm, slot in SLOTS:
                mhz, us = C.c_double(), C.c_double()
                if L.pdwt_clock_probe_read(slot, C.byref(mhz), C.byref(us)) == 0 and us.value > 0:
                    out[nm] = (mhz.value, us.value)
            return out
        pw0 = time.time()
        nprobe = max(5, min(steps, 30))
        acc = {}
        for _ in range(4):  # four sustained runs of nprobe back-to-back steps: the probe holds the LAST launch of each kind
            for _ in range(nprobe):
                step()
            sync()
            for nm, v in read_slots().items():
                acc.setdefault(nm, []).append(v)
        pw1 = time.time()
        L.pdwt_clock_probe_enable(0)
        clock_probe = {k: {"shader_mhz_median": round(sorted(m for m, _ in v)[len(v) // 2], 1), "shader_mhz_min": round(min(m for m, _ in v), 1),
                           "workgroup0_us_median": round(sorted(u for _, u in v)[len(v) // 2], 1), "samples": len(v)} for k, v in acc.items()}
        clock_probe["amdsmi_same_window"] = _SAMPLER.window(pw0, pw1) if _SAMPLER is not None else None
        clock_probe["what"] = ("s_memtime / s_memrealtime of workgroup 0 of the 8192^2 (level 1) and 4096^2 (level 2) launches: the last step of "
                               "each of four sustained back-to-back runs of %d steps" % nprobe)
    extra_timing = None
    if cfg["extra"]:
        # The timed step runs the drop-in C++ class's default (reference behaviour, src/wt.cu:398-418): norm1() reduces the bands every
        # time it is called (set_norm_cache(0), see where W is created).  The opt-in one-pass form -- soft_threshold() leaves sum|c| behind
        # for the norm1() that follows (the Python wrapper's own default) -- is timed here, next to the headline value.
        W.set_norm_cache(True)
        # (the same settle phase as the timed region: the sanity checks above left the GPU idle, and five warm-up steps after an idle
        #  spell understate what the one-pass form saves -- 20 us instead of ~100)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
            for _ in range(5):
                step()
            sync()
        t1 = time.perf_counter()
        nrep = max(5, min(steps, 20))
        for _ in range(nrep):
            step()
        torch.cuda.synchronize()
        extra_timing = {"norm_in_threshold_ms_per_step": round((time.perf_counter() - t1) / nrep * 1e3, 5),
                        "note": "same step with Wavelets::set_norm_cache(1) (opt-in): soft_threshold() accumulates sum|c| of what it writes and norm1() returns it; "
                                "ms_per_step / value above are the C++ class default, norm1() reducing the bands every time"}
        W.set_norm_cache(False)
    cpu = None
    if rank == 0 and world == 1 and cpu_seconds > 0:
        cpu = cpu_baseline(cfg, cpu_seconds)
    del W
    del Ws
    return {"value": round(value, 1), "unit": cfg["unit"], "ms_per_step": round(ms_per_step, 5), "gpu_ms_per_step": round(gpu_ms / steps, 5),
            "steps": steps, "warmup": warmup, "levels": levels_eff, "workload": cfg["desc"], "dtype": "f32" if cfg["dtype"] == "float32" else "f64",
            "sanity": sanity, "roundtrip_max_rel_err": rt_err, "roofline": roofline, "cpu_baseline": cpu, "power": power,
            "images_per_step": nbatch, "ms_per_image_pair": round(ms_per_step / nbatch, 5), "extra_timing": extra_timing, "clock_probe": clock_probe, "steady_state": steady,
            "kernels": {k: {a: round(b, 3) for a, b in v.items()} for k, v in kernels.items()}}


# ---------------------------------------------------------------------------------------------------------------------
# Shader clock and socket power during the timed region ("report sclk and power next to the number").  A helper
# PROCESS polls amdsmi's gpu_metrics every ~2 ms into a file (a thread of this process would compete with the launch loop for
# the interpreter lock); run_config() keeps the samples whose timestamps fall inside its timed region.  None when amdsmi is
# not importable on the host.
# ---------------------------------------------------------------------------------------------------------------------
_SAMPLER_SRC = r"""
import os, sys, time
path, idx = sys.argv[1], int(sys.argv[2])
parent = os.getppid()
out = open(path, "w", buffering=1)
try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[idx]
    lim = amdsmi.amdsmi_get_power_info(h).get("power_limit")
    out.write("# limit %s\n" % lim)
except Exception as e:
    out.write("# error %r\n" % (e,))
    sys.exit(0)
while os.getppid() == parent:  # (gone with the bench process, whatever happens to it)
    try:
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        ck = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
        clk = sum(ck) / len(ck) if ck else m.get("current_gfxclk")
        out.write("%.6f %s %s %s\n" % (time.time(), clk, m.get("current_socket_power"), m.get("temperature_hotspot")))
    except Exception as e:
        out.write("# error %r\n" % (e,))
    time.sleep(0.002)
"""


class PowerSampler:
    def __init__(self, device_index):
        import subprocess, tempfile
        self.path, self.proc = None, None
        try:
            fd, self.path = tempfile.mkstemp(prefix="pdwt_smi_", suffix=".txt")
            os.close(fd)
            self.proc = subprocess.Popen([sys.executable, "-c", _SAMPLER_SRC, self.path, str(device_index)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None

    def window(self, t0, t1):
        """Samples with t0 <= timestamp <= t1 -> summary dict (None when there are none)."""
        if not self.proc or not self.path:
            return None
        allrec, limit = [], None
        try:
            for ln in open(self.path):
                if ln.startswith("# limit"):
                    try:
                        limit = float(ln.split()[2]) / 1e6  # microwatts
                    except Exception:
                        pass
                    continue
                if ln.startswith("#"):
                    continue
                f = ln.split()
                if len(f) < 3:
                    continue
                try:
                    rec = (float(f[0]), float(f[1]), float(f[2]), float(f[3]))
                except Exception:
                    continue
                allrec.append(rec)
        except Exception:
            return None
        inside = [r for r in allrec if t0 <= r[0] <= t1]
        how = "inside the timed region"
        if not inside:  # a timed region shorter than the polling period: the samples right around it (same load: warm-up before, roofline pass after)
            inside = [r for r in allrec if t0 - 0.02 <= r[0] <= t1 + 0.02]
            how = "within 20 ms of the timed region (shorter than the polling period)"
        clk, pw, temp = [r[1] for r in inside], [r[2] for r in inside], [r[3] for r in inside]
        if not clk:
            return None
        return {"window": how,"sclk_mhz": {"mean": round(sum(clk) / len(clk), 0), "min": min(clk), "max": max(clk)},
                "socket_w": {"mean": round(sum(pw) / len(pw), 0), "max": max(pw)}, "power_limit_w": limit,
                "hotspot_c_max": max(temp) if temp else None, "samples": len(clk),
                "source": "amdsmi gpu_metrics (mean of the per-XCD gfx clocks, socket power), polled every ~2 ms by a helper process during the timed region"}

    def close(self):
        try:
            if self.proc:
                self.proc.kill()
                self.proc.wait(timeout=5)
            if self.path and os.path.exists(self.path):
                os.unlink(self.path)
        except Exception:
            pass


_SAMPLER = None


_PDWT = None


def pdwt_amd_mod():
    global _PDWT
    if _PDWT is None:
        import pdwt_amd
        _PDWT = pdwt_amd
    return _PDWT


# short timed runs of the other BASELINE configs appended to the headline line ("other_configs"): (steps, warmup, settle ms, cpu s)
OTHER_RUNS = {"c2_batch": (40, 5, 60.0, 0.0), "c3": (60, 10, 60.0, 5.0), "c4": (200, 20, 60.0, 5.0), "c5": (30, 5, 60.0, 6.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)  # >= ~0.1 s of GPU work: the clocks need ~50 ms of load to settle
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other configs (other_configs)")
    ap.add_argument("--batch-instances", action="store_true", help="c2_batch: cycle 16 private Wavelets instances (one image per launch) instead of the ImageBatch entry")
    ap.add_argument("--settle-ms", type=float, default=150.0, help="untimed load before the warm-up steps (clock settling)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under torch.distributed.run, RCCL over
            # xGMI for the barrier / max-over-ranks / norm all-reduce); rank 0 of the children prints the one JSON line
            import socket
            import subprocess
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            raise SystemExit(subprocess.call(cmd))
        args.gpus = world

    # torch first: its bundled ROCm runtime then also serves libpdwt_hip.so (one HIP runtime per process)
    import torch
    import torch.distributed as dist
    pdwt_amd = pdwt_amd_mod()

    # PDWT_BENCH_ONE_GPU=1 (test hook, with PDWT_BENCH_BACKEND=gloo): all ranks on device 0, so that the N > 1 code path of this
    # file can be exercised on a 1-GPU box (tests/test_batch_gpu.py); the driver's runs use one GPU per rank over RCCL
    if os.environ.get("PDWT_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    L = pdwt_amd.hip()
    assert L.pdwt_set_device(local_rank) == 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("PDWT_BENCH_BACKEND", "nccl") == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    global _SAMPLER
    # every rank samples the clock and socket power of ITS device (N > 1: eight GPUs drawing power at once is what a scaling run is about)
    _SAMPLER = PowerSampler(local_rank)
    time.sleep(0.5)  # (the helper imports amdsmi)
    # the size of the collective group, from the communicator itself: an all-reduce of one 1 per rank over the backend the batch split uses
    rccl_ranks, coll_backend = None, None
    if world > 1:
        coll_backend = dist.get_backend()
        one = torch.ones(1, device="cuda" if coll_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(one)
        rccl_ranks = int(round(float(one.item())))
    cfg = CONFIGS[args.config]
    res = run_config(args.config, args, L, torch, dist, rank, world, args.steps, args.warmup, args.settle_ms, args.cpu_seconds, not args.no_roofline)

    # The other BASELINE configs, short runs in the same process (N = 1 only; the headline fields above stay those of --config)
    others = None
    if world == 1 and not args.no_others and args.config == "c2":
        others = {}
        for name, (st, wu, settle, cpu_s) in OTHER_RUNS.items():
            try:
                r = run_config(name, args, L, torch, dist, rank, world, st, wu, settle, cpu_s if args.cpu_seconds > 0 else 0.0, not args.no_roofline)
                others[name] = r
            except Exception as e:  # a failing side run must not take the headline line with it -- but it must be visible
                others[name] = {"error": repr(e)}
            torch.cuda.empty_cache()
    # N > 1: the headline workload lives in every GPU's Infinity Cache, so its weak-scaling curve says little about N GPUs streaming from HBM
    # at once.  Every rank therefore also times, briefly, the two workloads that DO stream: c2_batch (16 distinct images per GPU through the
    # batched entry) and its C4 shard (8192 x 8192 rows of the batched-1D array) -- whole-job values, max over ranks, like the headline.
    streaming = None
    if world > 1 and not args.no_others and args.config == "c2":
        streaming = {}
        for name in ("c2_batch", "c4"):
            st, wu, settle, _ = OTHER_RUNS[name]
            try:
                r = run_config(name, args, L, torch, dist, rank, world, st, wu, settle, 0.0, False)
                streaming[name] = {k: r[k] for k in ("value", "unit", "ms_per_step", "gpu_ms_per_step", "ms_per_image_pair", "steps", "warmup", "workload", "sanity", "power")}
            except Exception as e:
                streaming[name] = {"error": repr(e)}
            torch.cuda.empty_cache()
    # per-rank clock / power means of the headline's timed region, gathered as numbers
    per_rank = None
    if world > 1:
        pw = res.get("power") or {}
        mine = torch.tensor([float((pw.get("sclk_mhz") or {}).get("mean") or -1.0), float((pw.get("socket_w") or {}).get("mean") or -1.0)],
                            device="cuda" if coll_backend == "nccl" else "cpu", dtype=torch.float64)
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        per_rank = {"sclk_mhz_mean": [None if float(t[0]) < 0 else float(t[0]) for t in allp],
                    "socket_w_mean": [None if float(t[1]) < 0 else float(t[1]) for t in allp],
                    "what": "amdsmi means of each rank's own device inside the headline's timed region (null: no amdsmi on that rank)"}

    if rank == 0:
        line = {
            "metric": "Mpixels/s fwd+inv DWT (4096\u00b2 db4 L3)" if args.config == "c2" else "%s fwd+inv (%s)" % (cfg["unit"], args.config),
            "value": res["value"], "unit": cfg["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": res["dtype"], "data": "synthetic",
            "config": {"workload": cfg["desc"], "images_per_gpu_per_step": res["images_per_step"], "levels": res["levels"],
                       "parallelism": "batch-split x%d (no data-path collective)" % world,
                       "working_set": ("one image per GPU, transformed over and over: its ~170 MB stay in the 256 MiB Infinity Cache; the same workload on a "
                                       "batch that streams through HBM is other_configs.c2_batch") if args.config == "c2" else None},
            "gpu_ms_per_step": res["gpu_ms_per_step"], "settle_ms": args.settle_ms, "roundtrip_max_rel_err": res["roundtrip_max_rel_err"],
            "sanity": res["sanity"], "roofline": res["roofline"], "cpu_baseline": res["cpu_baseline"], "power": res["power"],
            "extra_timing": res.get("extra_timing"), "clock_probe": res.get("clock_probe"), "steady_state": res.get("steady_state"),
            "kernels": res["kernels"], "other_configs": others,
            "rccl_ranks": rccl_ranks if coll_backend == "nccl" else None, "collective": {"backend": coll_backend, "ranks": rccl_ranks} if world > 1 else None,
            "per_rank": per_rank,
        }
        # the HBM-streaming throughputs next to the cache-resident headline, at every N: whole-job Mpixels/s of 16 distinct images per GPU
        # through the batched entry, and whole-job Msamples/s of the C4 shards (N = 1: the same numbers as other_configs.c2_batch / .c4)
        src = (streaming if world > 1 else others) or {}
        cbs, c4s = src.get("c2_batch") or {}, src.get("c4") or {}
        line["value_streaming"] = cbs.get("value")
        line["ms_per_image_pair_streaming"] = cbs.get("ms_per_image_pair")
        line["c4_value"] = c4s.get("value")
        line["c4_ms_per_step"] = c4s.get("ms_per_step")
        line["streaming_runs"] = streaming
        # both memory regimes of the headline workload at the top level of `roofline`: the figures above are one image living in the
        # Infinity Cache; `streaming` is the same transform on 16 distinct images in one batched call (pdwt_batch2d_*), 4.3 GB per
        # step: the number that is an HBM number
        cb = (others or {}).get("c2_batch")
        if line["roofline"] is not None and cb and cb.get("roofline"):
            rb = cb["roofline"]
            line["roofline"]["regime"] = "Infinity-Cache-resident (one image, ~170 MB working set)"
            line["roofline"]["streaming"] = {
                "what": "other_configs.c2_batch: 16 distinct 4096^2 images per step through the batched entry (ImageBatch / pdwt_batch2d_*), working set out of the Infinity Cache",
                "ms_per_image_pair": cb.get("ms_per_image_pair"), "Mpixels_per_s": cb.get("value"),
                "step_frac_of_peak": rb.get("step_frac_of_peak"), "kernel": rb.get("kernel"), "achieved": rb.get("achieved"), "frac": rb.get("frac"),
                "avg_launch_us": rb.get("avg_launch_us"), "copy_ceiling": rb.get("copy_ceiling")}
        print(json.dumps(line), flush=True)
    if _SAMPLER is not None:
        _SAMPLER.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
