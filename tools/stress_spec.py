#!/usr/bin/env python3
"""Randomised bit-exactness check of the cascade kernels in the neighbourhood of the geometries that have straight-line wave programs
(heights 3.8k-4.9k and 7.9k-8.6k rows, any width, filters of up to 8 taps, 2-4 levels) against one launch per level, single images and
image batches; prints how many cases really ran the wave programs.  usage: PYTHONPATH=. python tools/stress_spec.py [n] [seed]"""
import ctypes as C
import sys

import numpy as np
import torch  # noqa: F401
import pdwt_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = pdwt_amd.hip()


def stat(name):
    v = C.c_int()
    assert L.pdwt_debug_get(name, C.byref(v)) == 0
    return v.value


def run(x, wname, lev, casc):
    L.pdwt_debug_set(b"casc", casc)
    L.pdwt_debug_set(b"casc_min", 0)
    W = pdwt_amd.Wavelets(x, wname, lev)
    W.forward()
    c = W.coeffs
    W.inverse()
    out = (c, W.get_image())
    W.close()
    return out


bad = ran_f = ran_i = 0
for it in range(n):
    wname = ["db2", "db3", "db4", "sym4", "sym2", "coif1"][rs.randint(6)]
    tall = rs.randint(5) == 0
    nr = 8 * (rs.randint(988, 1075) if tall else rs.randint(475, 612))
    nc = 4 * rs.randint(128, 1300 if tall else 2200)
    lev = rs.randint(2, 5)
    x = rs.uniform(-100, 100, (nr, nc)).astype(np.float32)
    f0, i0 = stat(b"stat_casc_spec_fwd"), stat(b"stat_casc_spec_inv")
    a = run(x, wname, lev, 1)
    sf, si = stat(b"stat_casc_spec_fwd") > f0, stat(b"stat_casc_spec_inv") > i0
    ran_f += sf
    ran_i += si
    b = run(x, wname, lev, 0)
    L.pdwt_debug_set(b"casc", 1)
    ok = all(np.array_equal(p, q) for p, q in zip(a[0], b[0])) and np.array_equal(a[1], b[1])
    rt = np.abs(a[1] - x).max() / np.abs(x).max()
    if not ok or rt > 1e-5:
        bad += 1
        print("MISMATCH", wname, nr, nc, lev, rt, "spec fwd/inv", sf, si)

# image batches: the batched cascade entries against the single-image instance, bit for bit
nb_bad = 0
for it in range(max(2, n // 8)):
    wname = ["db2", "db4", "sym4"][rs.randint(3)]
    nr = 8 * rs.randint(475, 560)
    nc = 4 * rs.randint(256, 1200)
    lev = rs.randint(2, 4)
    nimg = rs.randint(2, 6)
    xs = rs.uniform(-50, 50, (nimg, nr, nc)).astype(np.float32)
    B = pdwt_amd.ImageBatch(xs, wname, lev)
    B.forward()
    cb = [B[i].coeffs for i in range(nimg)]
    B.inverse()
    ib = B.get_images()
    for i in range(nimg):
        c1, i1 = run(xs[i], wname, lev, 1)
        ok = np.array_equal(ib[i], i1)
        ok = ok and all(np.array_equal(p, q) for p, q in zip(cb[i], c1))
        if not ok:
            nb_bad += 1
            print("BATCH MISMATCH", wname, nr, nc, lev, nimg, i)
    B.close()
print("stress_spec: %d cases, %d bad; wave programs ran in %d forward / %d inverse cases; batch mismatches %d" % (n, bad, ran_f, ran_i, nb_bad))
sys.exit(1 if (bad or nb_bad) else 0)
