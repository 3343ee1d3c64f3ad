import numpy as np, torch, pdwt_amd, ctypes as C, sys
from tests.helpers import knobs
rs = np.random.RandomState(5)
names = ["db2","db3","db5","db6","db7","db9","db10","db11","db13","db14","db15","db17","db19","sym5","sym7","sym13","coif1","coif2","coif3","coif5","bior1.3","bior2.4","bior3.7","bior6.8","rbio3.9"]
bad = 0
for wn in names:
    for (nr, nc, lev) in ((1030, 516, 3), (256, 640, 2), (2048, 2048, 2)):
        x = rs.uniform(-10, 10, (nr, nc))
        res = []
        for kn in (dict(), dict(f64_lds=3)):
            with knobs(f64_lds_min=0, **kn):
                W = pdwt_amd.Wavelets(x, wn, lev)
                W.forward(); c = W.coeffs; W.inverse(); res.append((c, W.get_image()))
        ok = all(np.array_equal(a, b) for a, b in zip(res[0][0], res[1][0])) and np.array_equal(res[0][1], res[1][1])
        rt = float(np.abs(res[0][1] - x).max())
        if not ok or rt > 1e-9: bad += 1; print("MISMATCH", wn, nr, nc, lev, ok, rt)
print("padded-bank check:", len(names) * 3, "cases,", bad, "bad")
