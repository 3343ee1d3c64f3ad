#!/usr/bin/env python3
"""C5 transforms (8192^2 f64 db20 L6 forward + inverse) per kernel, for the library in PDWT_LIBDIR and the inverse workgroup target in
PDWT_EXP1 (0 = default).  One line per call; run the variants interleaved from a shell loop on one box."""
import ctypes as C, os, time, torch, pdwt_amd
L = pdwt_amd.hip()
x = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)
W = pdwt_amd.Wavelets(None, "db20", 6, dtype="float64", shape=(8192, 8192), device_ptr=x.data_ptr())
for _ in range(5):
    W.forward(); W.inverse()
W.sync()
t0 = time.perf_counter()
for _ in range(20):
    W.forward(); W.inverse()
W.sync()
dt = (time.perf_counter() - t0) / 20 * 1e3
L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
for _ in range(10):
    W.forward(); W.inverse()
W.sync()
n, ms = C.c_int(), C.c_double()
k = {}
for i in range(L.pdwt_kernel_count()):
    L.pdwt_ktime_read(i, C.byref(n), C.byref(ms))
    if n.value: k[L.pdwt_kernel_name(i).decode()] = round(ms.value * 1e3 / 10, 1)
W.forward(); W.inverse()
err = float((torch.as_tensor(W.get_image()) - x.cpu()).abs().max())
print("%-28s exp1=%-4s pair %.4f ms  %s  roundtrip %.1e" % (os.path.basename(os.environ.get("PDWT_LIBDIR", "lib")), os.environ.get("PDWT_EXP1", "0"), dt, k, err))
