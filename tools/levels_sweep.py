import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, pdwt_amd
L = pdwt_amd.hip()
x = torch.rand(4096, 4096, device="cuda") * 255
for wn in ("db4", "db2"):
    for lev in (2, 3, 4, 5):
        W = pdwt_amd.Wavelets(x, wn, lev)
        for _ in range(50):
            W.forward(); W.inverse()
        L.pdwt_sync()
        t0 = time.perf_counter()
        for _ in range(300):
            W.forward(); W.inverse()
        L.pdwt_sync()
        print(wn, "L%d" % lev, "%.2f us per pair" % ((time.perf_counter() - t0) / 300 * 1e6), "casc_l3 =", os.environ.get("PDWT_CASC_L3", "1"))
