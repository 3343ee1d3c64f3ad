#!/usr/bin/env python3
"""Bands past 2^31 elements: one SWT level of an N x N float32 image (every band has N*N samples); norm1 and the soft threshold
against torch on the band views, then the round trip.  usage: PYTHONPATH=. python tools/big_check_swt.py [N=46400] [wname=db2]"""
import sys

import torch
import pdwt_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 46400
wname = sys.argv[2] if len(sys.argv) > 2 else "db2"
g = torch.Generator(device="cuda").manual_seed(12)
x = torch.rand((N, N), generator=g, device="cuda", dtype=torch.float32)
W = pdwt_amd.Wavelets(x, wname, 1, do_swt=1)
assert W.state == pdwt_amd.W_INIT
W.forward()
W.sync()
b = [torch.as_tensor(W.coeff_view(k), device="cuda") for k in range(W.nbands)]
assert all(tuple(t.shape) == (N, N) for t in b), [t.shape for t in b]
ref = sum(float(t.abs().double().sum()) for t in b)
got = float(W.norm1())
print("norm1 %.9e vs torch %.9e  rel %.2e" % (got, ref, abs(got - ref) / ref))
assert abs(got - ref) / ref < 1e-6
W.inverse()
W.sync()
rec = torch.as_tensor(W.image_view(), device="cuda")
err = max(float((rec[r:r + 4096] - x[r:r + 4096]).abs().max()) for r in range(0, N, 4096))
print("round trip max abs error %.3e" % err)
assert err < 1e-5
W.forward()
W.sync()
beta = 0.05
tail = [t[-64:].clone() for t in b[1:]]  # the far end of each detail band (element offsets > 2^31)
W.soft_threshold(beta)
W.sync()
for t, old in zip(b[1:], tail):
    want = torch.sign(old) * torch.clamp(old.abs() - beta, min=0)
    assert torch.equal(t[-64:], want), float((t[-64:] - want).abs().max())
print("soft threshold at the far end of the detail bands: exact")
print("BIG SWT OK")
