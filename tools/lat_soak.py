#!/usr/bin/env python3
"""Run-to-run determinism of the hand-counted lattice level kernels (dwt_lat.hip) under load: an 8192^2 float64 db20 L2 transform (and the batched-1D
non-temporal-load kernels, 8192 x 8192 float32 sym8 L4) repeated many times must give bit-identical coefficients and reconstruction every time -- a
wait count one too small, or a compiler copy of an in-flight load register, would show up as rare differing digests.
usage: python tools/lat_soak.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pdwt_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
total_bad = 0
for what, shape, wname, lev, dt, ndim in (("lattice 2-D", (8192, 8192), "db20", 2, torch.float64, 2), ("lattice 2-D tall", (12288, 4096), "db20", 1, torch.float64, 2),
                                          ("batched 1-D nt", (8192, 8192), "sym8", 4, torch.float32, 1)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(shape, generator=g, device="cuda", dtype=dt) * 255 - 100
    torch.cuda.synchronize()
    W = pdwt_amd.Wavelets(x, wname, lev, ndim=ndim)
    it32 = torch.int32
    views, ref, bad = None, None, 0
    for it in range(reps):
        W.set_image(x)
        W.forward()
        W.sync()
        if views is None:
            views = [torch.as_tensor(W.coeff_view(k), device="cuda") for k in range(W.nbands)]
        sig = [int(v.view(it32).to(torch.int64).sum().item()) for v in views]
        W.inverse()
        W.sync()
        sig.append(int(torch.as_tensor(W.image_view(), device="cuda").view(it32).to(torch.int64).sum().item()))
        if ref is None:
            ref = sig
        elif sig != ref:
            bad += 1
            print(what, "iteration", it, "differs in", [i for i, (a, b) in enumerate(zip(sig, ref)) if a != b], flush=True)
    print("%s %s %s L%d: %d repetitions, %d differing" % (what, shape, wname, lev, reps, bad), flush=True)
    total_bad += bad
    del W, views
sys.exit(1 if total_bad else 0)
