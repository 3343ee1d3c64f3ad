#!/usr/bin/env python3
"""Fused SWT levels in double precision (swt_fused_f64.inc) against the two-pass kernels and the oracle, plus timing."""
import time
import numpy as np
import torch
import pdwt_amd
from tests.helpers import knobs
from oracle import oracle as orc
L = pdwt_amd.hip()
rs = np.random.RandomState(2)
for wn, shape, lev in (("db7", (512, 1024), 4), ("db2", (300, 640), 3), ("sym8", (1024, 512), 5), ("haar", (256, 512), 4), ("coif2", (640, 1030), 3)):
    x = rs.uniform(0, 255, shape)
    res = []
    for kn in (1, 0):
        with knobs(swtf_f64=kn):
            W = pdwt_amd.Wavelets(x, wn, lev, do_swt=1)
            W.forward(); c = W.coeffs; W.inverse(); res.append((c, W.get_image())); W.close()
    O = orc.OracleWavelets(x, wn, lev, do_swt=1)
    O.forward()
    eb = max(float(np.abs(g - o).max() / max(np.abs(o).max(), 1e-30)) for g, o in zip(res[0][0], O.coeffs))
    O.inverse()
    same_fwd = all(np.array_equal(a, b) for a, b in zip(res[0][0], res[1][0]))
    print(wn, shape, lev, "forward == two-pass:", same_fwd, " bands vs oracle %.2e  inverse vs oracle %.2e  vs two-pass %.2e" % (
        eb, float(np.abs(res[0][1] - O.get_image()).max() / 255), float(np.abs(res[0][1] - res[1][1]).max() / 255)), flush=True)
for kn in (1, 0):
    with knobs(swtf_f64=kn):
        x = torch.rand((4096, 4096), device="cuda", dtype=torch.float64)
        torch.cuda.synchronize()
        W = pdwt_amd.Wavelets(x, "db7", 3, do_swt=1)
        for _ in range(3): W.forward(); W.inverse()
        L.pdwt_sync(); t0 = time.perf_counter()
        for _ in range(10): W.forward(); W.inverse()
        L.pdwt_sync()
        print("swt f64 db7 L3 4096^2, fused =", kn, ": %.0f us per pair" % ((time.perf_counter() - t0) / 10 * 1e6), flush=True)
        W.close()
