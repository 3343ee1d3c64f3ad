"""Write-only and read-only streaming rates out of the Infinity Cache (torch kernels; GB/s of useful bytes)."""
import torch, time
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    bufs = [torch.empty(n, device="cuda") for _ in range(4 if mb < 4096 else 2)]
    i = [0]
    def fill():
        i[0] = (i[0] + 1) % len(bufs); bufs[i[0]].fill_(1.0)
    def rsum():
        i[0] = (i[0] + 1) % len(bufs); bufs[i[0]].sum()
    def copy():
        i[0] = (i[0] + 1) % len(bufs); bufs[i[0]].copy_(bufs[i[0] - 1])
    def w4():  # 1 read, 4 writes (the forward SWT level's pattern), through torch: 4 fills + 1 sum
        pass
    print("%5d MB: fill %.0f GB/s   sum(read) %.0f GB/s   copy %.0f GB/s (r+w)" % (mb, mb * 1.048576e6 / t(fill) / 1e9, mb * 1.048576e6 / t(rsum) / 1e9, 2 * mb * 1.048576e6 / t(copy) / 1e9))
