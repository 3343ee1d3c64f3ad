import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pdwt_amd

from tests.helpers import knobs, band_err
for shape, lev in (((4096,8192),2), ((8192,4096),2), ((16384,4096),1), ((4104,4096),1), ((12288,12288),2)):
    x = torch.rand(*shape, device="cuda", dtype=torch.float64)*200-100
    res={}
    for lat in (1,0):
        with knobs(f64_lat=lat):
            W = pdwt_amd.Wavelets(None, "db20", lev, dtype="float64", shape=shape, device_ptr=x.data_ptr())
            W.forward(); c=[W.get_coeff(k) for k in range(W.nbands)]
            W.inverse(); res[lat]=(c, W.get_image())
    xe = x.cpu().numpy()
    e = max(band_err(a,b) for a,b in zip(res[1][0],res[0][0]))
    print(shape, lev, "lattice vs direct bands %.2e  round trip lattice %.2e direct %.2e" % (e, band_err(res[1][1], xe), band_err(res[0][1], xe)), flush=True)
