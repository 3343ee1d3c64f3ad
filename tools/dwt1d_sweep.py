#!/usr/bin/env python3
"""Batched 1-D DWT (8192 rows x 8192 samples, L4) across bank lengths, float32 and float64: kernel time per direction and the
fraction of the FP pipe / of a copy that is.  PYTHONPATH=. python tools/dwt1d_sweep.py"""
import ctypes as C
import torch
import pdwt_amd

L = pdwt_amd.hip()
for dt, tdt in (("float32", torch.float32), ("float64", torch.float64)):
    x = torch.rand(8192, 8192, device="cuda", dtype=tdt)
    print("| %s wavelet (taps) | fwd us | inv us | pair: TB/s on compulsory bytes | GFMA/s |" % dt)
    print("|---|---|---|---|---|")
    for wname in ("db2", "db4", "sym8", "db10", "db12", "db16", "db20"):
        W = pdwt_amd.Wavelets(None, wname, 4, ndim=1, dtype=dt, shape=(8192, 8192), device_ptr=x.data_ptr())
        for _ in range(3):
            W.forward(); W.inverse()
        W.sync()
        L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
        reps = 5
        for _ in range(reps):
            W.forward(); W.inverse()
        W.sync()
        cnt, ms = C.c_int(), C.c_double()
        f = i = 0.0
        for k in range(L.pdwt_kernel_count()):
            L.pdwt_ktime_read(k, C.byref(cnt), C.byref(ms))
            if cnt.value:
                nm = L.pdwt_kernel_name(k).decode()
                if nm.startswith(("fwd", "ana")):
                    f += ms.value
                else:
                    i += ms.value
        L.pdwt_ktime_enable(0); L.pdwt_ktime_reset()
        hl = W.info.hlen
        es = 4 if dt == "float32" else 8
        byts = 4 * 8192 * 8192 * es
        fma = 2 * 8192 * 8192 * 1.875 * hl
        print("| %s (%d) | %.0f | %.0f | %.2f | %.0f |" % (wname, hl, f / reps * 1e3, i / reps * 1e3, byts / ((f + i) / reps * 1e-3) / 1e12, fma / ((f + i) / reps * 1e-3) / 1e9))
        del W
    del x
