#!/usr/bin/env python3
"""Batched 1-D in double precision at the C4 shape (8192 rows x 8192 samples): the one-buffer fused kernels of round 5 (knob dwt1d_f64 = 1)
against the per-level kernels (0), interleaved on one box; per-kernel event times next to the step.
   PYTHONPATH=. python tools/dwt1d_f64_ab.py"""
import ctypes as C
import time
import numpy as np
import torch
import pdwt_amd

L = pdwt_amd.hip()
for (nr, nc, wname, lev) in [(8192, 8192, "sym8", 4), (8192, 8192, "db4", 4), (8192, 8192, "db10", 4), (8192, 6144, "sym8", 3)]:
    x = torch.randn(nr, nc, device="cuda", dtype=torch.float64)
    res, outs, kern = {}, {}, {}
    for rep in range(2):
        for on in (1, 0):
            L.pdwt_debug_set(b"dwt1d_f64", on)
            W = pdwt_amd.Wavelets(None, wname, lev, ndim=1, dtype="float64", shape=(nr, nc), device_ptr=x.data_ptr())
            for _ in range(5):
                W.forward(); W.inverse()
            W.sync()
            t0 = time.perf_counter()
            for _ in range(30):
                W.forward(); W.inverse()
            W.sync()
            res.setdefault(on, []).append((time.perf_counter() - t0) / 30 * 1e3)
            if rep == 0:
                L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
                for _ in range(10):
                    W.forward(); W.inverse()
                W.sync()
                n, ms = C.c_int(), C.c_double()
                kern[on] = {}
                for k in range(L.pdwt_kernel_count()):
                    L.pdwt_ktime_read(k, C.byref(n), C.byref(ms))
                    if n.value:
                        kern[on][L.pdwt_kernel_name(k).decode()] = (n.value / 10, round(ms.value * 1e3 / 10, 1))
                L.pdwt_ktime_enable(0); L.pdwt_ktime_reset()
                W.forward()
                outs[on] = [c.copy() for c in W.coeffs]
                W.inverse()
                outs[on].append(W.get_image())
            del W
    same = all(np.array_equal(a, b) for a, b in zip(outs[1], outs[0]))
    byt = 4.0 * nr * nc * 8
    print("%5d x %5d f64 %s L%d: fused %s ms (%.2f TB/s on compulsory bytes), per-level %s ms, bit-identical %s\n   kernels fused %s\n   kernels per-level %s"
          % (nr, nc, wname, lev, ["%.3f" % v for v in res[1]], byt / (min(res[1]) * 1e-3) / 1e12, ["%.3f" % v for v in res[0]], same, kern[1], kern[0]))
L.pdwt_debug_set(b"dwt1d_f64", 1)
