#!/usr/bin/env python3
"""db20 / float64 2-D levels: the fused kernels against the two-pass kernels (bit for bit), plus forward / inverse times.
usage: PYTHONPATH=. python tools/f64_check.py [levels] size[xsize] ...   (knobs through PDWT_* as usual)"""
import ctypes as C
import sys
import time
import torch
import pdwt_amd

L = pdwt_amd.hip()
lev = int(sys.argv[1])
import os
WN = os.environ.get("WNAME", "db20")
DT = torch.float32 if os.environ.get("DTYPE", "f64") == "f32" else torch.float64


def run(x, lev, **kn):
    old = {}
    for k, v in kn.items():
        cur = C.c_int(0)
        assert L.pdwt_debug_get(k.encode(), C.byref(cur)) == 0
        old[k] = cur.value
        L.pdwt_debug_set(k.encode(), v)
    W = pdwt_amd.Wavelets(x.clone(), WN, lev)
    W.forward()
    W.sync()
    co = [torch.from_numpy(W.get_coeff(i)) for i in range(W.nbands)]
    W.inverse()
    img = torch.from_numpy(W.get_image())
    W.close()
    for k, v in old.items():
        L.pdwt_debug_set(k.encode(), v)
    return co, img


for a in sys.argv[2:]:
    nr, nc = (int(v) for v in a.split("x")) if "x" in a else (int(a), int(a))
    torch.manual_seed(nr * 7 + nc)
    x = torch.rand((nr, nc), device="cuda", dtype=DT) - 0.5
    c1, i1 = run(x, lev)
    c0, i0 = run(x, lev, force_twopass=1)
    bad = [k for k in range(len(c0)) if not torch.equal(c0[k], c1[k])]
    derr = max(float((p - q).abs().max()) for p, q in zip(c0, c1))
    print(WN, "%dx%d L%d  bands differing: %s (max |d| %.3g)  image equal: %s (max |d| %.3g)  roundtrip %.3g" % (
        nr, nc, lev, bad, derr, torch.equal(i0, i1), float((i0 - i1).abs().max()), float((i1 - x.cpu()).abs().max())), flush=True)
    W = pdwt_amd.Wavelets(x, WN, lev)
    reps = max(5, min(200, int(2e8 / (nr * nc))))
    for _ in range(3):
        W.forward(); W.inverse()
    L.pdwt_sync()
    out = []
    for which in ("fwd", "inv"):
        t0 = time.perf_counter()
        for _ in range(reps):
            if which == "fwd": W.forward()
            else: W.state = pdwt_amd.W_FORWARD; W.inverse()
        L.pdwt_sync()
        out.append((time.perf_counter() - t0) / reps * 1e6)
    print("          fwd %8.1f us   inv %8.1f us" % tuple(out), flush=True)
    W.close()
