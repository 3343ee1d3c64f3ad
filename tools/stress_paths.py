#!/usr/bin/env python3
"""Randomised parity sweep of the HIP paths against the CPU oracle (small random problems, every transform kind and both
precisions) -- a wider net than the fixed cases of tests/test_gpu_parity.py.  usage: PYTHONPATH=. python tools/stress_paths.py [n] [seed]"""
import sys

import numpy as np
import torch  # noqa: F401
import pdwt_amd
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
WN = ["haar", "db2", "db3", "db4", "db5", "db7", "db10", "sym4", "sym8", "coif2", "bior2.4", "bior3.3", "db20"]
bad = 0


def err(a, b):
    d = np.abs(b).max()
    return float(np.abs(a.astype(np.float64) - b).max() / (d if d > 0 else 1.0))


for it in range(n):
    kind = ["dwt2", "dwt1", "swt2", "swt1"][rs.randint(4)]
    dt = [np.float32, np.float64][rs.randint(2)]
    wname = WN[rs.randint(len(WN))]
    lev = rs.randint(1, 5)
    if kind == "dwt2":
        shape = (rs.randint(40, 400), rs.randint(40, 700))
    elif kind == "dwt1":
        shape = (rs.randint(1, 9), rs.randint(64, 3000))
    elif kind == "swt2":
        shape = (16 * rs.randint(4, 20), 16 * rs.randint(4, 40))
    else:
        shape = (rs.randint(1, 6), 16 * rs.randint(8, 120))
    kw = dict(do_swt=int(kind.startswith("swt")), ndim=1 if kind.endswith("1") else 2)
    x = rs.uniform(-100, 100, shape).astype(dt)
    W = pdwt_amd.Wavelets(x, wname, lev, **kw)
    if W.state == pdwt_amd.W_CREATION_ERROR:
        continue
    O = orc.OracleWavelets(x, wname, lev, **kw)
    W.forward()
    O.forward()
    tol = 1e-5 if dt == np.float32 else 1e-10
    e = max(err(g, o) for g, o in zip(W.coeffs, O.coeffs))
    beta = float(np.median(np.abs(W.get_coeff(W.nbands - 1))))
    W.soft_threshold(beta)
    O.soft_threshold(beta)
    n1 = abs(float(W.norm1()) - float(O.norm1())) / max(float(O.norm1()), 1e-30)
    W.inverse()
    O.inverse()
    e2 = err(W.get_image(), O.get_image())
    if e > tol or e2 > 4 * tol or n1 > (1e-5 if dt == np.float32 else 1e-10):
        bad += 1
        print("MISMATCH", kind, np.dtype(dt).name, wname, shape, lev, e, e2, n1)
print("stress_paths: %d cases, %d bad" % (n, bad))
sys.exit(1 if bad else 0)
