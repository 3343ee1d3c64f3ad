#!/usr/bin/env python3
"""fwd / inv time of the float32 2-D DWT per size, separately (events around each direction), for knob sweeps.
usage: PYTHONPATH=. python tools/size_sweep2.py wname levels size [size...]"""
import sys
import time
import torch
import pdwt_amd
wname, lev = sys.argv[1], int(sys.argv[2])
L = pdwt_amd.hip()
for n in [int(a) for a in sys.argv[3:]]:
    x = torch.rand((n, n), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    W = pdwt_amd.Wavelets(x, wname, lev)
    reps = max(20, min(1000, int(1e9 / (n * n * 16))))
    for _ in range(reps // 4 + 5):
        W.forward(); W.inverse()
    L.pdwt_sync()
    out = []
    for which in ("fwd", "inv", "pair"):
        t0 = time.perf_counter()
        for _ in range(reps):
            if which != "inv": W.forward()
            if which != "fwd": W.state = pdwt_amd.W_FORWARD; W.inverse()
        L.pdwt_sync()
        out.append((time.perf_counter() - t0) / reps * 1e6)
    print("%5d^2  fwd %7.1f  inv %7.1f  pair %7.1f us" % (n, *out), flush=True)
    W.close()
