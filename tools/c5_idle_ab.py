import ctypes as C, time, torch, pdwt_amd
L = pdwt_amd.hip()
x = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)
res = {}
for rep in range(3):
    for v in (0, 1):
        L.pdwt_debug_set(b"exp0", v)
        W = pdwt_amd.Wavelets(None, "db20", 6, dtype="float64", shape=(8192, 8192), device_ptr=x.data_ptr())
        for _ in range(3):
            W.forward(); W.inverse()
        W.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            W.forward(); W.inverse()
        W.sync()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
        for _ in range(5):
            W.forward(); W.inverse()
        W.sync()
        n, ms = C.c_int(), C.c_double()
        k = {}
        for i in range(L.pdwt_kernel_count()):
            L.pdwt_ktime_read(i, C.byref(n), C.byref(ms))
            if n.value: k[L.pdwt_kernel_name(i).decode()] = round(ms.value * 1e3 / 5, 1)
        L.pdwt_ktime_enable(0); L.pdwt_ktime_reset()
        W.forward(); c = [a.copy() for a in W.coeffs]; W.inverse(); img = W.get_image()
        res.setdefault(v, []).append((round(dt, 4), k, img))
        del W
import numpy as np
print("exp0=0 (broadcast idle lanes):", [(r[0], r[1]) for r in res[0]])
print("exp0=1 (round-4 form):        ", [(r[0], r[1]) for r in res[1]])
print("bit-identical:", np.array_equal(res[0][0][2], res[1][0][2]))
