#!/usr/bin/env python3
"""What a 20-step timed region (the driver's `bench.py --steps 20 --warmup 5`) sees against a long one: wall and event time per step of
repeated short regions in ONE process, with different idle gaps in front of the region.  PYTHONPATH=. python tools/k20_probe.py"""
import time
import torch
import pdwt_amd

L = pdwt_amd.hip()
x = torch.rand((4096, 4096), device="cuda", dtype=torch.float32) * 255
torch.cuda.synchronize()
W = pdwt_amd.Wavelets(None, "db4", 3, shape=(4096, 4096), dtype="float32", device_ptr=x.data_ptr())


def step():
    W.forward(); W.inverse()


import ctypes, os
HIP = ctypes.CDLL("libamdhip64.so")
HIP.hipStreamQuery.argtypes = [ctypes.c_void_p]
STREAM = L.pdwt_get_stream()
SPIN = os.environ.get("K20_SPIN", "0") == "1"


def sync():
    if SPIN:
        while HIP.hipStreamQuery(STREAM) != 0:  # hipErrorNotReady = 600
            pass
    L.pdwt_sync(); torch.cuda.synchronize()


def region(k, gap_us=0.0, spin=False):
    sync()
    if gap_us:
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < gap_us:
            pass
    e0, e1 = L.pdwt_event_create(), L.pdwt_event_create()
    t0 = time.perf_counter()
    L.pdwt_event_record(e0)
    for _ in range(k):
        step()
    L.pdwt_event_record(e1)
    tl = time.perf_counter()
    sync()
    t1 = time.perf_counter()
    g = L.pdwt_event_elapsed_ms(e0, e1)
    return (t1 - t0) / k * 1e6, g / k * 1e3, (tl - t0) / k * 1e6


t = time.perf_counter()
while time.perf_counter() - t < 0.15:
    for _ in range(20):
        step()
    sync()
print("steps, gap us: wall us/step, event us/step, host enqueue us/step")
for k, gap in ((20, 0), (20, 0), (20, 0), (20, 0), (20, 0), (20, 100), (20, 100), (20, 1000), (20, 1000), (20, 10000), (20, 10000), (20, 0), (20, 0),
               (100, 0), (100, 0), (2000, 0), (2000, 0), (20, 0), (20, 0), (5, 0), (5, 0)):
    w, g, h = region(k, gap)
    print("%5d %6d: %.2f %.2f %.2f" % (k, gap, w, g, h))
