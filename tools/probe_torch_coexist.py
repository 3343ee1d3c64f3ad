"""Probe: does libpdwt_hip.so work when torch (bundled ROCm runtime) is imported first?"""
import time
t0 = time.time()
import torch
print("torch import %.1fs" % (time.time() - t0), torch.__version__, torch.cuda.is_available(), torch.cuda.device_count())
import numpy as np
import pdwt_amd
x = np.random.RandomState(0).rand(512, 512).astype(np.float32)
W = pdwt_amd.Wavelets(x, "db4", 3)
W.forward(); W.inverse()
torch.cuda.synchronize()
print("roundtrip err", np.abs(W.get_image() - x).max())
t = torch.zeros(4, device="cuda") + 1
print("torch tensor ok", t.sum().item())
maps = open("/proc/self/maps").read()
print(sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l}))
# device pointer interop: a torch tensor handed to the library (memisonhost=0)
img = torch.rand(256, 256, device="cuda", dtype=torch.float32)
torch.cuda.synchronize()
W2 = pdwt_amd.Wavelets(None, "haar", 1, shape=(256, 256), device_ptr=img.data_ptr())
W2.forward(); W2.inverse()
print("interop err", np.abs(W2.get_image() - img.cpu().numpy()).max())
