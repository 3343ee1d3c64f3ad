#!/usr/bin/env python3
"""Per-workgroup start / end times of ONE launch of the fused level kernels of dwt_lds.hip (the C5 kernels), from their own
clocks (pdwt_clock_probe_enable(2 | 3)): who finishes when, by XCD and by dispatch order.
  python tools/lds_trace.py [--size 8192] [--wname db20] [--dir fwd|inv]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pdwt_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--wname", default="db20")
ap.add_argument("--dir", default="fwd")
ap.add_argument("--dtype", default="float64")
a = ap.parse_args()
import torch
L = pdwt_amd.hip()
x = torch.randn(a.size, a.size, device="cuda", dtype=torch.float64 if a.dtype == "float64" else torch.float32)
W = pdwt_amd.Wavelets(None, a.wname, 1, dtype=a.dtype, shape=(a.size, a.size), device_ptr=x.data_ptr())
for _ in range(10):
    W.forward()
    W.inverse()
W.sync()
L.pdwt_clock_probe_enable(2 if a.dir == "fwd" else 3)
for _ in range(5):
    W.forward()
    W.inverse()
W.sync()
NB = 4096
buf = np.zeros(NB * 4, dtype=np.uint64)
assert L.pdwt_clock_probe_dump(buf.ctypes.data_as(C.c_void_p), NB) == 0
L.pdwt_clock_probe_enable(0)
r = buf.reshape(NB, 4).astype(np.int64)
r = r[(r[:, 1] > 0) & (r[:, 3] > r[:, 1])]
t0 = r[:, 1].min()
st, en = (r[:, 1] - t0) / 100.0, (r[:, 3] - t0) / 100.0
mhz = (r[:, 2] - r[:, 0]) / (r[:, 3] - r[:, 1]) * 100.0
n = len(r)
print("%s level 1 of %dx%d %s: %d workgroups recorded" % (a.dir, a.size, a.size, a.wname, n))
q = lambda v, p: float(np.percentile(v, p))
print("start us: min %.1f  5%% %.1f  median %.1f  95%% %.1f  max %.1f" % (st.min(), q(st, 5), q(st, 50), q(st, 95), st.max()))
print("end   us: min %.1f  5%% %.1f  median %.1f  95%% %.1f  max %.1f" % (en.min(), q(en, 5), q(en, 50), q(en, 95), en.max()))
life = en - st
print("life  us: min %.1f  5%% %.1f  median %.1f  95%% %.1f  max %.1f" % (life.min(), q(life, 5), q(life, 50), q(life, 95), life.max()))
print("clock MHz: min %.0f median %.0f max %.0f" % (mhz.min(), q(mhz, 50), mhz.max()))
idx = np.arange(n)
for x8 in range(8):
    m = (idx % 8) == x8
    print("  XCD %d: start median %.1f  end median %.1f  max %.1f  life median %.1f" % (x8, q(st[m], 50), q(en[m], 50), en[m].max(), q(life[m], 50)))
late = st > 5.0
print("workgroups that start later than 5 us: %d (second round); their life median %.1f us" % (late.sum(), q(life[late], 50) if late.any() else 0))
# histogram of lifetimes and the slowest / fastest workgroups by dispatch index
h, edges = np.histogram(life, bins=12)
print("life histogram:", ", ".join("%.0f-%.0f:%d" % (edges[i], edges[i + 1], h[i]) for i in range(len(h))))
order = np.argsort(life)
ids = np.nonzero((buf.reshape(NB, 4)[:, 1] > 0))[0]
print("slowest 24 workgroups (dispatch index): ", sorted(ids[order[-24:]].tolist()))
print("fastest 24 workgroups (dispatch index): ", sorted(ids[order[:24]].tolist()))
# same-CU pairs cannot be identified from here; mean life by dispatch index modulo 64 and by index // 64
for mod in (16, 32, 64):
    print("life by (index mod %d):" % mod, " ".join("%.0f" % life[(ids % mod) == k].mean() for k in range(mod)))
print("life by (index // 32):", " ".join("%.0f" % life[(ids // 32) == k].mean() for k in range((ids.max() // 32) + 1)))
