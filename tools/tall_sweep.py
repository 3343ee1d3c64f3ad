#!/usr/bin/env python3
"""Tall images: one workgroup per CU (loop-form cascade kernels, casc_spec = 3) against several rounds of workgroups of the C2 height with
wave programs in the forward (casc_spec = 7), interleaved on one box.  PYTHONPATH=. python tools/tall_sweep.py"""
import time, ctypes as C
import torch, pdwt_amd
L = pdwt_amd.hip()
for n in (8192, 16384, 6144, 12288, 8192):
    x = torch.rand((n, n), device="cuda", dtype=torch.float32); torch.cuda.synchronize()
    W = pdwt_amd.Wavelets(x, "db4", 3)
    res = {3: [], 7: []}
    for rep in range(3):
        for spec in (3, 7):
            L.pdwt_debug_set(b"casc_spec", spec)
            for _ in range(5): W.forward(); W.inverse()
            L.pdwt_sync(); reps = 30; t0 = time.perf_counter()
            for _ in range(reps): W.forward(); W.inverse()
            L.pdwt_sync(); res[spec].append((time.perf_counter() - t0) / reps * 1e6)
    print(n, "one round: %s us   several rounds: %s us" % (["%.1f" % v for v in res[3]], ["%.1f" % v for v in res[7]]))
    W.close(); del x
L.pdwt_debug_set(b"casc_spec", 7)
