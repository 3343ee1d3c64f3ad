#!/usr/bin/env python3
"""Forward+inverse time of less common transform kinds (which kernels does a reference user land on?).
usage: PYTHONPATH=. python tools/corner_sweep.py"""
import time
import numpy as np
import torch
import pdwt_amd
L = pdwt_amd.hip()
cases = [("2d f32 db4 L3 4095x4097", dict(shape=(4095, 4097), wname="db4", levels=3, dtype=torch.float32)),
         ("1d f32 sym8 L4 8192x8192", dict(shape=(8192, 8192), wname="sym8", levels=4, dtype=torch.float32, ndim=1)),
         ("1d f32 db16 L4 8192x8192", dict(shape=(8192, 8192), wname="db16", levels=4, dtype=torch.float32, ndim=1)),
         ("1d f32 db20 L4 8192x8192", dict(shape=(8192, 8192), wname="db20", levels=4, dtype=torch.float32, ndim=1)),
         ("1d f64 sym8 L4 8192x8192", dict(shape=(8192, 8192), wname="sym8", levels=4, dtype=torch.float64, ndim=1)),
         ("1d f64 db20 L4 8192x8192", dict(shape=(8192, 8192), wname="db20", levels=4, dtype=torch.float64, ndim=1)),
         ("swt f32 db7 L5 4096^2", dict(shape=(4096, 4096), wname="db7", levels=5, dtype=torch.float32, do_swt=1)),
         ("swt f32 db16 L3 4096^2", dict(shape=(4096, 4096), wname="db16", levels=3, dtype=torch.float32, do_swt=1)),
         ("swt f64 db7 L3 4096^2", dict(shape=(4096, 4096), wname="db7", levels=3, dtype=torch.float64, do_swt=1)),
         ("2d f64 db4 L3 4096^2", dict(shape=(4096, 4096), wname="db4", levels=3, dtype=torch.float64)),
         ("2d f32 bior6.8 L3 4096^2", dict(shape=(4096, 4096), wname="bior6.8", levels=3, dtype=torch.float32)),
         ("2d f32 haar L3 4096^2", dict(shape=(4096, 4096), wname="haar", levels=3, dtype=torch.float32))]
for name, c in cases:
    x = torch.rand(c["shape"], device="cuda", dtype=c["dtype"])
    torch.cuda.synchronize()
    W = pdwt_amd.Wavelets(x, c["wname"], c["levels"], ndim=c.get("ndim", 2), do_swt=c.get("do_swt", 0))
    for _ in range(3):
        W.forward(); W.inverse()
    L.pdwt_sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        W.forward(); W.inverse()
    L.pdwt_sync()
    dt = (time.perf_counter() - t0) / reps
    nbytes = x.numel() * x.element_size()
    print("%-28s %9.1f us per pair   %6.2f TB/s on 4x the image bytes" % (name, dt * 1e6, 4 * nbytes / dt / 1e12), flush=True)
    W.close()
