#!/usr/bin/env python3
"""What the instrumentation inside a 20-step timed bracket costs: HIP events around the loop, one synchronize or two.  PYTHONPATH=. python tools/k20_bracket.py"""
import time
import torch
import pdwt_amd

import ctypes, os
if os.environ.get("K20_SCHED"):
    HIP = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", HIP.hipSetDeviceFlags(int(os.environ["K20_SCHED"])))  # 1 spin, 2 yield, 4 blocking sync
L = pdwt_amd.hip()
x = torch.rand((4096, 4096), device="cuda", dtype=torch.float32) * 255
torch.cuda.synchronize()
W = pdwt_amd.Wavelets(None, "db4", 3, shape=(4096, 4096), dtype="float32", device_ptr=x.data_ptr())


def step():
    W.forward(); W.inverse()


def region(k, events, two_syncs):
    L.pdwt_sync(); torch.cuda.synchronize()
    if events:
        e0, e1 = L.pdwt_event_create(), L.pdwt_event_create()
    t0 = time.perf_counter()
    if events:
        L.pdwt_event_record(e0)
    for _ in range(k):
        step()
    if events:
        L.pdwt_event_record(e1)
    if two_syncs:
        L.pdwt_sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


t = time.perf_counter()
while time.perf_counter() - t < 0.15:
    for _ in range(20):
        step()
    torch.cuda.synchronize()
res = {}
for rep in range(12):
    for ev, ts in ((1, 1), (0, 1), (0, 0), (1, 0)):
        res.setdefault((ev, ts), []).append(region(20, ev, ts))
for k, v in res.items():
    v.sort()
    print("events=%d two_syncs=%d: median %.2f  min %.2f us per step (K=20)" % (k[0], k[1], v[len(v) // 2], v[0]))
print("K=2000: %.2f" % region(2000, 0, 0))
