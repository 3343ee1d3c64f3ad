#!/usr/bin/env python3
"""Randomised bit-exactness sweep of the kernels added in round 5 against the kernels they replace (knob off):
  * batched 1-D in double precision, rows whose two-buffer footprint exceeds the LDS budget (dwt1d_f64),
  * fused SWT levels for float32 banks of 22 ... 40 taps (swtf_long; inverse: 1e-5),
  * the batched 2-D entry in double precision against per-image transforms,
  * the LDS-tiled kernels of custom non-separable banks (nonsep_tiled = 2) against the one-thread-per-output kernels.
usage: PYTHONPATH=. python tools/stress_r5.py [n] [seed]"""
import ctypes as C
import sys

import numpy as np
import torch  # noqa: F401
import pdwt_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = pdwt_amd.hip()
bad = 0


def knob(name, v):
    assert L.pdwt_debug_set(name.encode(), int(v)) == 0


def err(a, b):
    d = np.abs(b).max()
    return float(np.abs(a.astype(np.float64) - b).max() / (d if d > 0 else 1.0))


W1 = ["db2", "db3", "db4", "sym5", "db7", "sym8", "db10", "bior3.5", "coif3", "db12", "db16", "db20"]
for it in range(n):
    # ---- 1-D double, long rows
    nc = int(rs.choice([8192, 8190, 7168, 6146, 5120, 4610, 7000, 8000]))
    nr = int(rs.randint(1, 700))
    wname = W1[rs.randint(len(W1))]
    lev = int(rs.randint(1, 7))
    x = rs.randn(nr, nc)
    out = []
    for on in (1, 0):
        knob("dwt1d_f64", on)
        W = pdwt_amd.Wavelets(x, wname, lev, ndim=1)
        W.forward()
        c = W.coeffs
        W.inverse()
        out.append((c, W.get_image()))
    knob("dwt1d_f64", 1)
    ok = all(np.array_equal(a, b) for a, b in zip(out[0][0], out[1][0])) and np.array_equal(out[0][1], out[1][1]) and err(out[0][1], x) < 1e-9
    if not ok:
        bad += 1
        print("MISMATCH 1d f64", nr, nc, wname, lev)
W2 = ["db11", "db12", "db13", "db14", "db15", "db16", "db17", "db18", "db19", "db20", "sym11", "sym16", "sym20"]
for it in range(n):
    # ---- SWT long banks
    nr = int(rs.choice([192, 256, 320, 384, 512, 640]))
    nc = int(4 * rs.randint(40, 520))
    wname = W2[rs.randint(len(W2))]
    lev = int(rs.randint(1, 4))
    x = rs.uniform(-100, 100, (nr, nc)).astype(np.float32)
    res = []
    for on in (1, 0):
        knob("swtf_long", on)
        W = pdwt_amd.Wavelets(x, wname, lev, do_swt=1)
        W.forward()
        c = W.coeffs
        W.inverse()
        res.append((c, W.get_image(), W.info.nlevels))
    knob("swtf_long", 1)
    ok = all(np.array_equal(a, b) for a, b in zip(res[0][0], res[1][0])) and err(res[0][1], res[1][1]) < 1e-5 and err(res[0][1], x) < 1e-5
    if not ok:
        bad += 1
        print("MISMATCH swt long", nr, nc, wname, lev, res[0][2], [bool(np.array_equal(a, b)) for a, b in zip(res[0][0], res[1][0])], err(res[0][1], res[1][1]))
W3 = ["db2", "db4", "db5", "sym8", "db10", "db16", "db20", "bior4.4"]
for it in range(max(4, n // 2)):
    # ---- batched 2-D double
    B = int(rs.randint(2, 9))
    nr, nc = int(rs.randint(64, 600)), int(rs.randint(64, 600))
    wname = W3[rs.randint(len(W3))]
    lev = int(rs.randint(1, 4))
    x = rs.randn(B, nr, nc)
    IB = pdwt_amd.ImageBatch(x, wname, lev)
    IB.forward()
    ok = True
    singles = []
    for b in range(B):
        W = pdwt_amd.Wavelets(x[b], wname, lev)
        W.forward()
        ok = ok and all(np.array_equal(a, c) for a, c in zip(IB[b].coeffs, W.coeffs))
        singles.append(W)
    IB.inverse()
    o = IB.get_images()
    for b in range(B):
        singles[b].inverse()
        ok = ok and np.array_equal(o[b], singles[b].get_image())
    if not ok:
        bad += 1
        print("MISMATCH batch f64", B, nr, nc, wname, lev, IB.batched)
for it in range(n):
    # ---- custom non-separable banks: LDS-tiled kernels (every size: knob 2) against the one-thread-per-output kernels
    dt = (np.float32, np.float64)[rs.randint(2)]
    swt = int(rs.randint(3) == 0)
    hl = int(rs.randint(1, 41)) if rs.randint(4) == 0 else int(rs.randint(1, 13))
    nr, nc = int(rs.randint(hl + 8, 700)), int(rs.randint(hl + 8, 900))
    if swt:
        nr, nc = (nr + 7) // 8 * 8, (nc + 7) // 8 * 8
    lev = int(rs.randint(1, 4))
    kf = [rs.randn(hl, hl) for _ in range(4)]
    ki = [rs.randn(hl, hl) for _ in range(4)]
    x = rs.uniform(-5, 5, (nr, nc)).astype(dt)
    res = []
    for tiled in (2, 0):
        knob("nonsep_tiled", tiled)
        W = pdwt_amd.Wavelets(x, "db2", lev, do_separable=0, do_swt=swt)
        assert W.set_filters_forward_nonseparable("custom2d", *kf) == 0 and W.set_filters_inverse_nonseparable(*ki) == 0
        W.forward()
        c = W.coeffs
        W.inverse()
        res.append((c, W.get_image(), W.info.nlevels))
    knob("nonsep_tiled", 1)
    ok = res[0][2] == res[1][2] and all(np.array_equal(a, b) for a, b in zip(res[0][0], res[1][0])) and np.array_equal(res[0][1], res[1][1])
    if not ok:
        bad += 1
        print("MISMATCH nonsep", np.dtype(dt).name, swt, hl, nr, nc, lev)
print("stress_r5: %d cases per family, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
