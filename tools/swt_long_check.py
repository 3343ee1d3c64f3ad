import time, numpy as np, torch, pdwt_amd
from tests.helpers import knobs
from oracle import oracle as orc
L = pdwt_amd.hip()
for wn in ("db7", "db8", "db9", "db10", "db11"):
    x = torch.rand((4096, 4096), device="cuda")
    torch.cuda.synchronize()
    W = pdwt_amd.Wavelets(x, wn, 3, do_swt=1)
    for _ in range(3): W.forward(); W.inverse()
    L.pdwt_sync(); t0 = time.perf_counter()
    for _ in range(10): W.forward(); W.inverse()
    L.pdwt_sync(); dt = (time.perf_counter() - t0) / 10
    print(wn, "swt L3 4096^2: %.0f us per pair" % (dt * 1e6), flush=True)
    W.close()
rs = np.random.RandomState(1)
for wn in ("db9", "db10", "coif3"):
    xs = rs.uniform(0, 255, (512, 1024)).astype(np.float32)
    W = pdwt_amd.Wavelets(xs, wn, 3, do_swt=1); O = orc.OracleWavelets(xs, wn, 3, do_swt=1)
    W.forward(); O.forward()
    e = max(float(np.abs(g.astype(np.float64) - o).max() / np.abs(o).max()) for g, o in zip(W.coeffs, O.coeffs))
    W.inverse(); O.inverse()
    e2 = float(np.abs(W.get_image().astype(np.float64) - O.get_image()).max() / 255)
    print(wn, "vs oracle: bands %.2e inverse %.2e" % (e, e2), flush=True)
