#!/usr/bin/env python3
"""fwd+inv throughput of the float32 2-D DWT over image sizes (image resident in HBM).  usage: PYTHONPATH=. python tools/size_sweep.py [wname] [levels]"""
import sys
import time

import torch
import pdwt_amd

wname = sys.argv[1] if len(sys.argv) > 1 else "db4"
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = pdwt_amd.hip()
import os
print("PDWT_GRAPH =", os.environ.get("PDWT_GRAPH", "0"))
print("| size | us per fwd+inv | Mpixels/s | compulsory GB/s |")
print("|---|---|---|---|")
for n in (256, 512, 1024, 2048, 4096, 8192, 16384):
    x = torch.rand((n, n), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    W = pdwt_amd.Wavelets(x, wname, lev)
    reps = max(20, min(2000, int(2e9 / (n * n * 16))))
    for _ in range(10):
        W.forward(); W.inverse()
    L.pdwt_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        W.forward(); W.inverse()
    L.pdwt_sync()
    dt = (time.perf_counter() - t0) / reps
    print("| %d² | %.1f | %.0f | %.0f |" % (n, dt * 1e6, n * n / dt / 1e6, 16.0 * n * n / dt / 1e9))
    W.close()
