#!/usr/bin/env python3
"""Launch-bound regime: B small images as B instances (6 launches per image and pair) vs ONE ImageBatch (6 launches per batch),
against a device-to-device copy of the same byte volume.   python tools/batch2d_bench.py [--B 64] [--size 512]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import pdwt_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--wname", default="db4")
ap.add_argument("--levels", type=int, default=3)
ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
a = ap.parse_args()
x = torch.rand(a.B, a.size, a.size, device="cuda", dtype=torch.float64 if a.dtype == "float64" else torch.float32) * 255
L = pdwt_amd.hip()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    L.pdwt_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    L.pdwt_sync()
    return (time.perf_counter() - t0) / reps * 1e6


Ws = [pdwt_amd.Wavelets(x[b], a.wname, a.levels) for b in range(a.B)]
IB = pdwt_amd.ImageBatch(x, a.wname, a.levels)


def singles():
    for W in Ws:
        W.forward()
        W.inverse()


def batch():
    IB.forward()
    IB.inverse()


t_s, t_b = timed(singles), timed(batch)
nbytes = a.B * a.size * a.size * (8 if a.dtype == "float64" else 4)
y = torch.empty_like(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    y.copy_(x)
e0.record()
for _ in range(20):
    y.copy_(x)
    x.copy_(y)
e1.record()
torch.cuda.synchronize()
t_copy = e0.elapsed_time(e1) / 20 * 1e3  # us for 4 * nbytes moved (the pair's compulsory volume: read + write, both directions)
pix = a.B * a.size * a.size
print("%d x %dx%d %s L%d pairs: instances %.1f us (%.2f us per image, %.0f Mpix/s) | ImageBatch (batched=%s) %.1f us (%.2f us per image, %.0f Mpix/s) | "
      "copy of the same volume %.1f us -> batch at %.2f of the copy rate, %.1fx the per-image loop"
      % (a.B, a.size, a.size, a.wname, a.levels, t_s, t_s / a.B, pix / t_s, IB.batched, t_b, t_b / a.B, pix / t_b, t_copy, t_copy / t_b, t_s / t_b))
