#!/usr/bin/env python3
"""Randomised bit-exactness check of the two-levels-per-launch kernels against one launch per level
(many shapes x filter lengths).  usage: PYTHONPATH=. PDWT_CASC_MIN=0 python tools/stress_cascade.py [n] [seed]"""
import os
import sys

import numpy as np
import torch  # noqa: F401
import pdwt_amd

os.environ.setdefault("PDWT_CASC_MIN", "0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = pdwt_amd.hip()
bad = 0
KEY = os.environ.get("STRESS_KEY", "casc").encode()
for it in range(n):
    wname = ["db2", "db3", "db4", "db5", "db6", "db7", "db8", "sym4", "coif2"][rs.randint(9)]
    nr = 4 * rs.randint(160, 700)
    nc = 4 * rs.randint(64, 700)
    lev = rs.randint(2, 5)
    x = rs.uniform(-100, 100, (nr, nc)).astype(np.float32)
    res = []
    for casc in (1, 0):  # (key = casc or stream)
        L.pdwt_debug_set(KEY, casc)
        W = pdwt_amd.Wavelets(x, wname, lev)
        W.forward()
        c = W.coeffs
        W.inverse()
        res.append((c, W.get_image()))
    L.pdwt_debug_set(KEY, 1)
    ok = all(np.array_equal(a, b) for a, b in zip(res[0][0], res[1][0])) and np.array_equal(res[0][1], res[1][1])
    rt = np.abs(res[0][1] - x).max() / np.abs(x).max()
    if not ok or rt > 1e-5:
        bad += 1
        print("MISMATCH", wname, nr, nc, lev, rt)
print("stress_cascade: %d cases, %d bad" % (n, bad))
sys.exit(1 if bad else 0)
