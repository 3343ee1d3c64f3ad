import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, pdwt_amd
L = pdwt_amd.hip()
x = torch.rand(4096, 4096, device="cuda") * 255
W = pdwt_amd.Wavelets(x, "db4", 3)
for _ in range(200):
    W.forward(); W.inverse()
L.pdwt_sync()
N = 3000
t0 = time.perf_counter()
for _ in range(N):
    W.forward(); W.inverse()
t1 = time.perf_counter()
L.pdwt_sync()
t2 = time.perf_counter()
print("host enqueue %.2f us per step; total %.2f us per step (GPU-bound if enqueue < total)" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
# small image: host cost only
xs = torch.rand(256, 256, device="cuda")
Ws = pdwt_amd.Wavelets(xs, "db4", 3)
for _ in range(200):
    Ws.forward(); Ws.inverse()
L.pdwt_sync()
t0 = time.perf_counter()
for _ in range(N):
    Ws.forward(); Ws.inverse()
t1 = time.perf_counter()
L.pdwt_sync()
t2 = time.perf_counter()
print("256^2: host enqueue %.2f us per step; total %.2f" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
