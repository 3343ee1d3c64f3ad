#!/usr/bin/env python3
"""Run-to-run determinism of the asm-pipelined kernels under load: the same 4096^2 transform repeated many times must give
bit-identical coefficients and reconstruction every time (a wait count that is one too small, or a compiler copy of an
in-flight register, would show up as rare bit flips).  usage: PYTHONPATH=. python tools/determinism.py [reps] [wname]"""
import sys

import numpy as np
import torch
import pdwt_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((4096, 4096), generator=g, device="cuda", dtype=torch.float32) * 255
torch.cuda.synchronize()
W = pdwt_amd.Wavelets(x, wname, 3)
views = None
ref = None
bad = 0
for it in range(reps):
    W.set_image(x)
    W.forward()
    W.sync()
    if views is None:
        views = [torch.as_tensor(W.coeff_view(k), device="cuda") for k in range(W.nbands)]
    sig = [int(v.view(torch.int32).to(torch.int64).sum().item()) for v in views]
    W.inverse()
    W.sync()
    sig.append(int(torch.as_tensor(W.image_view(), device="cuda").view(torch.int32).to(torch.int64).sum().item()))
    if ref is None:
        ref = sig
    elif sig != ref:
        bad += 1
        print("iteration", it, "differs in", [i for i, (a, b) in enumerate(zip(sig, ref)) if a != b])
print("determinism: %d repetitions, %d differing" % (reps, bad))
sys.exit(1 if bad else 0)
