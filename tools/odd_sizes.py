import time, torch, pdwt_amd
L = pdwt_amd.hip()
for shape in [(4096,4096),(4094,4098),(4092,4100),(4095,4096),(4096,4095),(4095,4097),(2047,2049),(1000,1000),(1001,1003)]:
    x = torch.rand(shape, device="cuda")
    torch.cuda.synchronize()
    W = pdwt_amd.Wavelets(x, "db4", 3)
    for _ in range(5): W.forward(); W.inverse()
    L.pdwt_sync(); reps=50; t0=time.perf_counter()
    for _ in range(reps): W.forward(); W.inverse()
    L.pdwt_sync(); dt=(time.perf_counter()-t0)/reps
    L.pdwt_ktime_enable(1); L.pdwt_ktime_reset()
    W.forward(); W.inverse(); L.pdwt_sync()
    import ctypes as C
    ks=[]
    n, ms = C.c_int(), C.c_double()
    for k in range(L.pdwt_kernel_count()):
        L.pdwt_ktime_read(k, C.byref(n), C.byref(ms))
        if n.value: ks.append("%s x%d %.0fus" % (L.pdwt_kernel_name(k).decode(), n.value, ms.value*1e3))
    L.pdwt_ktime_enable(0)
    print(shape, "%.1f us/pair" % (dt*1e6), ks, flush=True)
    W.close()
