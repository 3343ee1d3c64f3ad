import numpy as np, torch, pdwt_amd
from tests.helpers import knobs
rs = np.random.RandomState(3)
bad = 0
for wn, shape, lev in (("db7", (4096, 4096), 5), ("db4", (2048, 3072), 5), ("sym8", (1024, 4096), 4), ("haar", (2048, 2048), 5), ("coif2", (4096, 1024), 5)):
    x = rs.uniform(0, 255, shape).astype(np.float32)
    res = []
    for pm in (1, 0):
        with knobs(swtf_perm=pm):
            W = pdwt_amd.Wavelets(x, wn, lev, do_swt=1)
            W.forward(); W.inverse(); res.append(W.get_image()); W.close()
    ok = np.array_equal(res[0], res[1])
    print(wn, shape, lev, "identical" if ok else "DIFFERENT", float(np.abs(res[0] - x).max() / 255))
    bad += (not ok)
print("bad", bad)
