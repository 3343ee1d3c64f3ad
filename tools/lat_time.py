#!/usr/bin/env python3
"""per-kernel times of an 8192^2 float64 db20 transform of L levels (argv: L [L ...]) for the library in PDWT_LIBDIR; PDWT_F64_LAT=0: lattice kernels off"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pdwt_amd
L = pdwt_amd.hip()
x = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)
for lev in [int(a) for a in sys.argv[1:]] or [1, 2, 6]:
    W = pdwt_amd.Wavelets(None, "db20", lev, dtype="float64", shape=(8192, 8192), device_ptr=x.data_ptr())
    for _ in range(5):
        W.forward(); W.inverse()
    W.sync()
    L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
    for _ in range(10):
        W.forward(); W.inverse()
    W.sync()
    n, ms = C.c_int(), C.c_double()
    k = {}
    for i in range(L.pdwt_kernel_count()):
        L.pdwt_ktime_read(i, C.byref(n), C.byref(ms))
        if n.value: k[L.pdwt_kernel_name(i).decode()] = round(ms.value * 1e3 / 10, 1)
    L.pdwt_ktime_enable(0)
    clk = {}
    if hasattr(L, "pdwt_clock_probe_enable"):
        L.pdwt_clock_probe_enable(1)
        for _ in range(20):
            W.forward(); W.inverse()
        W.sync()
        for nm, slot in (("f1", 1), ("i1", 9), ("f2", 2), ("i2", 10)):
            mhz, us = C.c_double(), C.c_double()
            if L.pdwt_clock_probe_read(slot, C.byref(mhz), C.byref(us)) == 0 and us.value > 0:
                clk[nm] = (round(mhz.value), round(us.value, 1))
        L.pdwt_clock_probe_enable(0)
    print("%-14s f64_lat=%s L%d %s clock(MHz, wg0 us) %s" % (os.path.basename(os.environ.get("PDWT_LIBDIR", "lib")), os.environ.get("PDWT_F64_LAT", "1"), lev, k, clk), flush=True)
