#!/usr/bin/env python3
"""Custom non-separable banks: the LDS-tiled kernels against the one-thread-per-output kernels (knob nonsep_tiled), interleaved on one box.
PYTHONPATH=. python tools/nonsep_ab.py"""
import time
import numpy as np
import torch
import pdwt_amd

L = pdwt_amd.hip()
print("| transform | size | kernel | levels | plain us (fwd / inv) | tiled us (fwd / inv) | ratio |")
print("|---|---|---|---|---|---|---|")
for dt, n, hl, lev, swt in ((np.float32, 4096, 8, 3, 0), (np.float32, 4096, 5, 3, 0), (np.float32, 2048, 16, 2, 0), (np.float64, 4096, 8, 3, 0),
                            (np.float64, 2048, 16, 2, 0), (np.float32, 2048, 8, 3, 1), (np.float32, 2048, 4, 4, 1), (np.float64, 2048, 8, 3, 1),
                            (np.float32, 512, 8, 3, 0)):
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.uniform(-1, 1, (n, n)).astype(dt)).cuda()
    kf = [rs.randn(hl, hl) for _ in range(4)]
    ki = [rs.randn(hl, hl) for _ in range(4)]
    W = pdwt_amd.Wavelets(x, "db2", lev, do_separable=0, do_swt=swt)
    assert W.set_filters_forward_nonseparable("custom2d", *kf) == 0 and W.set_filters_inverse_nonseparable(*ki) == 0
    res = {0: [], 1: []}
    for rep in range(2):
        for tiled in (0, 1):
            L.pdwt_debug_set(b"nonsep_tiled", tiled)
            t = []
            for fn in (W.forward, W.inverse):
                # (inverse() needs a forward() first each time: time forward alone, then the pair, and subtract)
                pass
            W.forward(); W.inverse(); L.pdwt_sync()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                W.forward()
            L.pdwt_sync()
            tf = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                W.forward(); W.inverse()
            L.pdwt_sync()
            tp = (time.perf_counter() - t0) / reps
            res[tiled].append((tf * 1e6, (tp - tf) * 1e6))
    L.pdwt_debug_set(b"nonsep_tiled", 1)
    p = min(res[0], key=lambda a: a[0] + a[1]); q = min(res[1], key=lambda a: a[0] + a[1])
    print("| %s | %d² %s | %dx%d | %d | %.0f / %.0f | %.0f / %.0f | %.1fx |" % ("SWT" if swt else "DWT", n, np.dtype(dt).name, hl, hl, lev, p[0], p[1], q[0], q[1], (p[0] + p[1]) / (q[0] + q[1])))
    W.close(); del x
