#!/usr/bin/env python3
"""Forward / inverse kernel time (HIP events per launch, summed per direction) of the float32 2-D DWT at 4096^2 L3 across bank lengths and
workgroup shapes of the cascade kernels (casc_wg / casc_iwg = waves per workgroup; 0 = the default) and with one launch per level (casc = 0).
PYTHONPATH=. python tools/casc_hlen_sweep.py [size]"""
import ctypes as C
import sys
import torch
import pdwt_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = pdwt_amd.hip()
x = torch.rand(n, n, device="cuda") * 255


def knob(k, v):
    assert L.pdwt_debug_set(k.encode(), int(v)) == 0


def run(wname, cfg):
    for k, v in (("casc", 1), ("casc_wg", 0), ("casc_iwg", 0)):
        knob(k, v)
    for k, v in cfg.items():
        knob(k, v)
    W = pdwt_amd.Wavelets(None, wname, 3, dtype="float32", shape=(n, n), device_ptr=x.data_ptr())
    for _ in range(5):
        W.forward(); W.inverse()
    W.sync()
    L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
    reps = 20
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
    del W
    return hl, f / reps * 1e3, i / reps * 1e3


print("%d^2 L3, us per direction (kernel time)" % n)
print("| wavelet (taps) | fwd: default | wg 16 | wg 8 | wg 4 | casc 0 | inv: default | iwg 16 | iwg 8 | iwg 4 | casc 0 |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for wname in ("db2", "db3", "db4", "db5", "db6", "db7", "db8", "sym8", "coif2", "bior4.4", "db9", "coif3", "db10"):
    fr, ir = [], []
    for cfg in ({}, {"casc_wg": 16, "casc_iwg": 16}, {"casc_wg": 8, "casc_iwg": 8}, {"casc_wg": 4, "casc_iwg": 4}, {"casc": 0}):
        hl, f, i = run(wname, cfg)
        fr.append("%.1f" % f); ir.append("%.1f" % i)
    print("| %s (%d) | %s | %s |" % (wname, hl, " | ".join(fr), " | ".join(ir)))
for k, v in (("casc", 1), ("casc_wg", 0), ("casc_iwg", 0)):
    knob(k, v)
