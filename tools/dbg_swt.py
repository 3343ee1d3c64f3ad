#!/usr/bin/env python3
"""debug: LDS-DMA fused SWT kernels vs the register-staged ones; prints where the bands differ."""
import sys
import numpy as np
import pdwt_amd
from tests.helpers import knobs
nr, nc, lev = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
wname = sys.argv[4] if len(sys.argv) > 4 else "db7"
x = np.random.RandomState(0).uniform(-1, 1, (nr, nc)).astype(np.float32)
res = []
for dma in (0, 1):
    with knobs(swtf_dma=dma):
        W = pdwt_amd.Wavelets(x, wname, lev, do_swt=1)
        W.forward()
        c = W.coeffs
        W.inverse()
        res.append((c, W.get_image()))
for k, (a, b) in enumerate(zip(res[0][0], res[1][0])):
    d = np.abs(a - b)
    bad = ~(d < 1e-5)
    if bad.any():
        rows = np.where(bad.any(axis=1))[0]
        cols = np.where(bad.any(axis=0))[0]
        print("band", k, "bad", bad.sum(), "rows", rows[:12], "... n=%d" % len(rows), "cols", cols[:12], "... n=%d" % len(cols), "nan", np.isnan(b).sum())
    else:
        print("band", k, "ok")
d = np.abs(res[0][1] - res[1][1])
print("inverse max diff", np.nanmax(d), "nan", np.isnan(res[1][1]).sum())
