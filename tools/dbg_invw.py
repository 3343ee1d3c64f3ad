#!/usr/bin/env python3
"""debug: workgroup-form inverse cascade vs the independent-wave one; prints which output rows / columns differ."""
import sys
import numpy as np
import pdwt_amd
from tests.helpers import knobs
nr, nc, lev = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wname = sys.argv[4] if len(sys.argv) > 4 else "db4"
iwg = int(sys.argv[5]) if len(sys.argv) > 5 else 8
rs = np.random.RandomState(0)
x = rs.uniform(-1, 1, (nr, nc)).astype(np.float32)
res = []
for kn in (dict(casc_iwg=1, casc_l3=0), dict(casc_iwg=iwg, casc_l3=0), dict(casc_iwg=iwg, casc_l3=1)):
    with knobs(casc_min=0, **kn):
        W = pdwt_amd.Wavelets(x, wname, lev)
        W.forward()
        W.inverse()
        res.append(W.get_image())
for name, r in (("wg", res[1]), ("wg+l3", res[2])):
    d = np.abs(r - res[0])
    bad = d > 1e-4
    rows = np.where(bad.any(axis=1))[0]
    cols = np.where(bad.any(axis=0))[0]
    print(name, "max err", d.max(), "bad rows", len(rows), "bad cols", len(cols))
    if len(rows):
        print("  rows:", rows[:40], "..." , rows[-10:])
        print("  cols:", cols[:40], "...", cols[-10:])
        print("  rows mod 4 hist", np.bincount(rows % 4, minlength=4), " bad per row (first):", bad[rows[0]].sum())
ref, got = res[0], res[1]
for dy in range(-12, 13):
    for dx in (-4, -2, 0, 2, 4):
        e = np.abs(np.roll(got, (dy, dx), axis=(0, 1)) - ref)
        frac = (e < 1e-4).mean()
        if frac > 0.05:
            print("shift", dy, dx, "matching fraction", frac)
print("ref[0:3,0:6]", ref[0:3, 0:6]); print("got[0:3,0:6]", got[0:3, 0:6]); print("x  [0:3,0:6]", x[0:3, 0:6])
print("is got finite", np.isfinite(got).all(), "got==0 frac", (got == 0).mean())
