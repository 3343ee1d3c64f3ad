#!/usr/bin/env python3
"""What the memory side of this box delivers for pure reads, pure writes and copies, inside and outside the 256 MiB
Infinity Cache (torch kernels, events): the ceilings the cascade kernels' phases are compared with."""
import torch

def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

for mb in (16, 64, 128, 512, 2048):
    n = mb * (1 << 20) // 4
    x = torch.rand(n, device="cuda")
    y = torch.empty_like(x)
    t_r = timeit(lambda: x.sum())
    t_r2 = timeit(lambda: torch.max(x))
    t_w = timeit(lambda: y.fill_(1.0))
    t_c = timeit(lambda: y.copy_(x))
    def wr_then_rd():
        y.fill_(2.0)
        return y.sum()
    t_wr = timeit(wr_then_rd)
    B = n * 4
    print("%5d MB: read(sum) %.2f us = %.2f TB/s | read(max) %.2f us = %.2f TB/s | write(fill) %.2f us = %.2f TB/s | copy %.2f us = %.2f TB/s (r+w) | fill then sum %.2f us (sum of parts %.2f)"
          % (mb, t_r, B / t_r / 1e6, t_r2, B / t_r2 / 1e6, t_w, B / t_w / 1e6, t_c, 2 * B / t_c / 1e6, t_wr, t_r + t_w))
