#!/usr/bin/env python3
"""Device-to-device copy rate against the working-set size (read + write bytes / time): what a pure stream reaches on this box
when it does / does not fit the 256 MB Infinity Cache.  usage: python tools/copy_ceiling.py"""
import torch
for mb in (32, 64, 128, 256, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.rand(n, device="cuda"); b = torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(10, min(400, 40000 // mb))
    e0.record()
    for _ in range(reps): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("copy %5d MB -> %5d MB moved: %8.1f us  %.2f TB/s" % (mb, 2 * mb, us, 2 * mb * 1.048576 / us))
