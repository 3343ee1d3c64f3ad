#!/usr/bin/env python3
"""SWT with float32 banks of 22 ... 40 taps at 4096^2: the two-columns-per-thread fused level kernels of round 5 (knob swtf_long = 1) against
the two-pass kernels (0), interleaved on one box.   PYTHONPATH=. python tools/swt_long_ab.py"""
import ctypes as C
import time
import torch
import pdwt_amd

L = pdwt_amd.hip()
x = torch.rand(4096, 4096, device="cuda") * 255
for wname, lev in (("db11", 3), ("db12", 3), ("db16", 3), ("db20", 3), ("db16", 5), ("db10", 3)):
    res, kern = {}, {}
    for rep in range(2):
        for on in (1, 0):
            L.pdwt_debug_set(b"swtf_long", on)
            W = pdwt_amd.Wavelets(None, wname, lev, do_swt=1, dtype="float32", shape=(4096, 4096), device_ptr=x.data_ptr())
            for _ in range(3):
                W.forward(); W.inverse()
            W.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                W.forward(); W.inverse()
            W.sync()
            res.setdefault(on, []).append((time.perf_counter() - t0) / 10 * 1e3)
            if rep == 0:
                L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
                for _ in range(5):
                    W.forward(); W.inverse()
                W.sync()
                n, ms = C.c_int(), C.c_double()
                kern[on] = {}
                for k in range(L.pdwt_kernel_count()):
                    L.pdwt_ktime_read(k, C.byref(n), C.byref(ms))
                    if n.value:
                        kern[on][L.pdwt_kernel_name(k).decode()] = (n.value / 5, round(ms.value * 1e3 / 5, 1))
                L.pdwt_ktime_enable(0); L.pdwt_ktime_reset()
            del W
    print("4096^2 f32 %s SWT L%d pair: fused %s ms, two-pass %s ms\n    fused %s\n    two-pass %s" % (wname, lev, ["%.3f" % v for v in res[1]], ["%.3f" % v for v in res[0]], kern[1], kern[0]))
L.pdwt_debug_set(b"swtf_long", 1)
