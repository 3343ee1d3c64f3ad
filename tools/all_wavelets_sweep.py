#!/usr/bin/env python3
"""Forward+inverse time of every named wavelet at one size (default 4096^2 float32 L3, decimated; --swt for the undecimated transform at 2048^2):
outliers show bank lengths that fall on a slow path.  PYTHONPATH=. python tools/all_wavelets_sweep.py [--swt] [--f64]"""
import sys
import time
import torch
import pdwt_amd

swt = "--swt" in sys.argv
f64 = "--f64" in sys.argv
n = 2048 if swt else 4096
L = pdwt_amd.hip()
L.pdwt_wavelet_name.restype = __import__("ctypes").c_char_p
x = torch.rand(n, n, device="cuda", dtype=torch.float64 if f64 else torch.float32) * 255
rows = []
for i in range(L.pdwt_num_wavelets()):
    w = L.pdwt_wavelet_name(i).decode()
    W = pdwt_amd.Wavelets(None, w, 3, do_swt=int(swt), dtype="float64" if f64 else "float32", shape=(n, n), device_ptr=x.data_ptr())
    for _ in range(3):
        W.forward(); W.inverse()
    W.sync()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        W.forward(); W.inverse()
    W.sync()
    rows.append((W.info.hlen, (time.perf_counter() - t0) / reps * 1e6, w))
    del W
by = {}
for h, t, w in rows:
    by.setdefault(h, []).append((t, w))
print("%d^2 %s %s L3: taps -> us per pair (min .. max over the wavelets of that length)" % (n, "f64" if f64 else "f32", "SWT" if swt else "DWT"))
for h in sorted(by):
    ts = sorted(by[h])
    print("  %2d taps: %7.1f .. %7.1f   (%d wavelets; slowest %s)" % (h, ts[0][0], ts[-1][0], len(ts), ts[-1][1]))
