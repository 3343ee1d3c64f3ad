#!/usr/bin/env python3
"""dwt_lat.hip against the oracle and against the direct-form level kernels (knob exp2 = 1 switches the lattice kernels off): parity on a few
shapes, then the per-kernel times of the C5 transform (8192^2 float64 db20 L6) with and without them."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pdwt_amd
from oracle import oracle as orc
from tests.helpers import knobs, band_err
L = pdwt_amd.hip()
orc.set_num_threads(orc.usable_cores())
wn = sys.argv[1] if len(sys.argv) > 1 else "db20"
for shape, lev in (((1024, 1024), 1), ((2048, 1024), 1), ((1024, 2048), 2), ((4096, 4096), 3)):
    rs = np.random.RandomState(shape[0] + lev)
    x = rs.uniform(-100, 100, shape)
    O = orc.OracleWavelets(x, wn, lev)
    O.forward()
    for off in (0, 1):
        with knobs(f64_lat=1 - off, f64_lat_min=512):
            W = pdwt_amd.Wavelets(x, wn, lev, dtype="float64")
            W.forward()
            ef = max(band_err(W.get_coeff(k), O.get_coeff(k)) for k in range(W.nbands))
            worst = int(np.argmax([band_err(W.get_coeff(k), O.get_coeff(k)) for k in range(W.nbands)]))
            W.inverse()
            ei = band_err(W.get_image(), x)
            # inverse alone on the ORACLE's coefficients
            W2 = pdwt_amd.Wavelets(x, wn, lev, dtype="float64")
            W2.forward()
            for k in range(W2.nbands):
                W2.set_coeff(O.get_coeff(k), k)
            W2.inverse()
            ei2 = band_err(W2.get_image(), x)
        print("%-12s L%d lattice %s: forward vs oracle %.2e (worst band %d)  round trip %.2e  inverse of oracle bands %.2e" % (shape, lev, "off" if off else "on ", ef, worst, ei, ei2), flush=True)
if "--time" in sys.argv:
    x = torch.randn(8192, 8192, device="cuda", dtype=torch.float64)
    for off in (1, 0, 1, 0):
        with knobs(f64_lat=1 - off):
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
            L.pdwt_ktime_enable(0)
            W.forward(); W.inverse()
            err = float((torch.as_tensor(W.get_image()) - x.cpu()).abs().max())
            print("lattice %s: pair %.4f ms  %s  roundtrip %.1e" % ("off" if off else "on ", dt, k, err), flush=True)
