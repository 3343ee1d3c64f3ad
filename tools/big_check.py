#!/usr/bin/env python3
"""Images past 2^31 elements / 2^32 bytes: the 2-D DWT on an N x N float32 image held only in HBM, checked by
  (1) level-1 detail coefficients at sampled positions (corners, the far end of the buffers, random) against the defining
      double sum over the wrapped hlen x hlen input patch (float64, SURVEY A-1), and
  (2) the forward -> inverse round trip over the whole image.
usage: PYTHONPATH=. python tools/big_check.py [N=46400] [wname=db4] [levels=3] [float32|float64] [ndim=2]
(ndim = 1: N rows of N samples, batched 1-D transform, D1 checked instead)"""
import sys

import numpy as np
import torch
import pdwt_amd
from oracle import oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 46400
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
lev = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dt = np.dtype(sys.argv[4] if len(sys.argv) > 4 else "float32")
ndim = int(sys.argv[5]) if len(sys.argv) > 5 else 2
tdt = torch.float32 if dt == np.float32 else torch.float64
tol = 2e-5 if dt == np.float32 else 1e-12
print("N = %d: %.3f G elements (2^31 = 2.147 G), %.2f GB per image" % (N, N * N / 1e9, N * N * dt.itemsize / 1e9))
g = torch.Generator(device="cuda").manual_seed(11)
x = torch.rand((N, N), generator=g, device="cuda", dtype=tdt)
torch.cuda.synchronize()
W = pdwt_amd.Wavelets(x, wname, lev, ndim=ndim)
assert W.state == pdwt_amd.W_INIT
W.forward()
W.sync()
hlen, _, F = orc.filters(wname, dt.type, 0)
L = np.array(F.L[:hlen], dtype=np.float64)
H = np.array(F.H[:hlen], dtype=np.float64)
c = hlen // 2 - 1
n2 = N // 2
rs = np.random.RandomState(3)
worst = 0.0
if ndim == 2:
    bands = [torch.as_tensor(W.coeff_view(k), device="cuda") for k in (1, 2, 3)]
    assert all(tuple(b.shape) == (n2, n2) for b in bands)
    pos = [(0, 0), (n2 - 1, n2 - 1), (n2 - 1, 0), (0, n2 - 1), (n2 - 2, n2 - 3)] + [(int(rs.randint(n2)), int(rs.randint(n2))) for _ in range(40)]
    assign = None
    for (i, j) in pos:
        rows = torch.tensor([(2 * i - c + k) % N for k in range(hlen)], device="cuda")
        cols = torch.tensor([(2 * j - c + k) % N for k in range(hlen)], device="cuda")
        patch = x[rows][:, cols].double().cpu().numpy()            # patch[a, b] = x[row a, col b]
        fr = {"L": L[::-1], "H": H[::-1]}                          # out = sum_k x[src(k)] F[hlen-1-k]
        cand = {rc: float(fr[rc[1]] @ patch @ fr[rc[0]]) for rc in ("LH", "HL", "HH")}  # rc = (filter along the row, filter down the column)
        got = [float(b[i, j]) for b in bands]
        if assign is None:  # which of bands 1, 2 is (row L, column H): fixed by the first sample, then required everywhere
            assign = ("LH", "HL", "HH") if abs(got[0] - cand["LH"]) <= abs(got[0] - cand["HL"]) else ("HL", "LH", "HH")
        for k in range(3):
            e = abs(got[k] - cand[assign[k]]) / max(1.0, abs(cand[assign[k]]))
            worst = max(worst, e)
            assert e < tol, ("band", k + 1, "at", (i, j), got[k], cand[assign[k]])
    print("level-1 details at %d positions (band order %s): worst relative error %.2e" % (len(pos), assign, worst))
else:
    d1 = torch.as_tensor(W.coeff_view(1), device="cuda")
    assert tuple(d1.shape) == (N, n2)
    pos = [(0, 0), (N - 1, n2 - 1), (N - 1, 0), (0, n2 - 1)] + [(int(rs.randint(N)), int(rs.randint(n2))) for _ in range(40)]
    for (r, j) in pos:
        cols = torch.tensor([(2 * j - c + k) % N for k in range(hlen)], device="cuda")
        want = float(x[r][cols].double().cpu().numpy() @ H[::-1])
        e = abs(float(d1[r, j]) - want) / max(1.0, abs(want))
        worst = max(worst, e)
        assert e < tol, ("D1 at", (r, j), float(d1[r, j]), want)
    print("D1 at %d positions: worst relative error %.2e" % (len(pos), worst))
W.inverse()
W.sync()
rec = torch.as_tensor(W.image_view(), device="cuda")
err = 0.0
for r0 in range(0, N, 4096):
    err = max(err, float((rec[r0:r0 + 4096] - x[r0:r0 + 4096]).abs().max()))
print("round trip max abs error %.3e (input in [0,1))" % err)
assert err < (1e-5 if dt == np.float32 else 1e-12)
print("BIG OK")
