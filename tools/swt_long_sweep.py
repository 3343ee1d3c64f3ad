import time, torch, pdwt_amd, sys
L = pdwt_amd.hip()
x = torch.rand(4096, 4096, device="cuda") * 255
for wname in ("db16", "db12", "db20"):
    for knobname, vals in ((b"swtf_mi", (0, 32, 48, 96, 128, 256)), (b"swtf_m", (0, 32, 48, 96, 128))):
        out = []
        for v in vals:
            L.pdwt_debug_set(knobname, v)
            W = pdwt_amd.Wavelets(None, wname, 3, do_swt=1, dtype="float32", shape=(4096, 4096), device_ptr=x.data_ptr())
            for _ in range(3):
                W.forward(); W.inverse()
            W.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                W.forward(); W.inverse()
            W.sync()
            out.append("%d: %.3f" % (v, (time.perf_counter() - t0) / 10 * 1e3))
            del W
        L.pdwt_debug_set(knobname, 0)
        print(wname, knobname.decode(), "  ".join(out))
