#!/usr/bin/env python3
"""SWT with float32 banks of 22 ... 40 taps at 4096^2 on the library set PDWT_LIBDIR points at: ms per forward+inverse pair and the kernel
times of one pair.  A/B of two builds on one box:  for d in lib_prev lib; do PDWT_LIBDIR=$PWD/pdwt_amd/$d python tools/swt_long_libs.py; done"""
import ctypes as C
import os
import time
import torch
import pdwt_amd

L = pdwt_amd.hip()
x = torch.rand(4096, 4096, device="cuda") * 255
print("libs:", os.environ.get("PDWT_LIBDIR", "pdwt_amd/lib"))
for wname, lev in (("db11", 3), ("db16", 3), ("db20", 3), ("db16", 5)):
    W = pdwt_amd.Wavelets(None, wname, lev, do_swt=1, dtype="float32", shape=(4096, 4096), device_ptr=x.data_ptr())
    for _ in range(3):
        W.forward(); W.inverse()
    W.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        W.forward(); W.inverse()
    W.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
    for _ in range(5):
        W.forward(); W.inverse()
    W.sync()
    n, t = C.c_int(), C.c_double()
    kern = {}
    for k in range(L.pdwt_kernel_count()):
        L.pdwt_ktime_read(k, C.byref(n), C.byref(t))
        if n.value:
            kern[L.pdwt_kernel_name(k).decode()] = (n.value / 5, round(t.value * 1e3 / 5, 1))
    L.pdwt_ktime_enable(0); L.pdwt_ktime_reset()
    print("  %s L%d pair: %.3f ms   %s" % (wname, lev, ms, kern))
    del W
