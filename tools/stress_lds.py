#!/usr/bin/env python3
"""Randomised bit-exactness of the LDS-ring level kernels (dwt_lds.hip) against the kernels they replace: random wavelets (all 72),
both precisions, random shapes (odd, even, not multiples of 4), random depths.  usage: PYTHONPATH=. python tools/stress_lds.py [n] [seed]"""
import sys
import ctypes as C
import numpy as np
import torch  # noqa: F401
import pdwt_amd
from tests.helpers import knobs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = pdwt_amd.hip()
names = [L.pdwt_wavelet_name(i).decode() for i in range(L.pdwt_num_wavelets())]
bad = 0
print("wavelets:", len(names), flush=True)
for it in range(n):
    if it % 50 == 0:
        print("case", it, file=sys.stderr, flush=True)
    wn = names[rs.randint(len(names))]
    dt = np.float32 if rs.randint(2) else np.float64
    nr, nc = int(rs.randint(90, 1500)), int(rs.randint(90, 1500))
    if rs.randint(3) == 0:
        nr, nc = 2 * (nr // 2), 2 * (nc // 2)
    lev = int(rs.randint(1, 5))
    x = rs.uniform(-50, 50, (nr, nc)).astype(dt)
    res = []
    for kn in (dict(), dict(f64_lds=0)):
        with knobs(f64_lds_min=0, **kn):
            W = pdwt_amd.Wavelets(x, wn, lev)
            if W.state != pdwt_amd.W_INIT:
                res = None
                break
            W.forward()
            c = W.coeffs
            W.inverse()
            res.append((c, W.get_image()))
            W.close()
    if res is None:
        continue
    ok = all(np.array_equal(a, b) for a, b in zip(res[0][0], res[1][0])) and np.array_equal(res[0][1], res[1][1])
    rt = float(np.abs(res[0][1].astype(np.float64) - x).max() / 50)
    tol = 2e-5 if dt == np.float32 else 1e-9
    if not ok or rt > tol:
        bad += 1
        print("MISMATCH", wn, dt.__name__, nr, nc, lev, ok, rt, flush=True)
print("stress_lds: %d cases, %d bad" % (n, bad), flush=True)
print("stress_lds: %d cases, %d bad" % (n, bad), file=sys.stderr, flush=True)
