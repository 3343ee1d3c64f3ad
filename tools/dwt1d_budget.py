#!/usr/bin/env python3
"""Batched 1-D, all levels in one launch vs one launch per level for rows whose LDS buffers exceed 80 KB (one workgroup per CU):
   PYTHONPATH=. python tools/dwt1d_budget.py   -> ms per forward+inverse pair with dwt1d_lds_kb = 80 (per-level kernels) and 158."""
import time
import numpy as np
import torch
import pdwt_amd

L = pdwt_amd.hip()
for (nr, nc, dt, wname, lev) in [(8192, 8192, "float64", "sym8", 4), (4096, 16384, "float32", "sym8", 4), (2048, 32768, "float32", "db4", 5), (8192, 8192, "float64", "db20", 3)]:
    x = torch.randn(nr, nc, device="cuda", dtype=torch.float64 if dt == "float64" else torch.float32)
    res, outs = {}, {}
    for kb in (80, 158):
        L.pdwt_debug_set(b"dwt1d_lds_kb", kb)
        W = pdwt_amd.Wavelets(None, wname, lev, ndim=1, dtype=dt, shape=(nr, nc), device_ptr=x.data_ptr())
        for _ in range(3):
            W.forward(); W.inverse()
        W.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            W.forward(); W.inverse()
        W.sync()
        res[kb] = (time.perf_counter() - t0) / 20 * 1e3
        W.forward()
        outs[kb] = [c.copy() for c in W.coeffs]
        W.inverse()
        del W
    same = all(np.array_equal(a, b) for a, b in zip(outs[80], outs[158]))
    print("%5d x %5d %s %s L%d: per-level %.3f ms, fused %.3f ms, bit-identical %s" % (nr, nc, dt, wname, lev, res[80], res[158], same))
L.pdwt_debug_set(b"dwt1d_lds_kb", 158)
