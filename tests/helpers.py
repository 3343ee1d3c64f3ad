"""Shared helpers for the parity tests."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# BASELINE.json north_star: "within 1e-5 relative" for float32; f64 bar from BASELINE.md section 3.
TOL = {np.dtype(np.float32): 1e-5, np.dtype(np.float64): 1e-10}


def band_err(a, b):
    """Band-normalised max error  max|a-b| / max|b|  (the metric of SURVEY.md 8c / BASELINE.md 3)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = np.abs(b).max()
    return np.abs(a - b).max() / (den if den > 0 else 1.0)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    for k in ("wname", "kind"):
        if k in d:
            d[k] = str(d[k])
    for k in ("levels", "nbands"):
        if k in d:
            d[k] = int(d[k])
    return d


def golden_bands(d):
    return [d["band%d" % i] for i in range(d["nbands"])]


class knobs:
    """with knobs(casc=0, casc_min=0): ...   -- set tuning knobs of libpdwt_hip.so (pdwt_debug_set), restore on exit."""

    def __init__(self, **kw):
        self.kw = kw
        self.old = {}

    def __enter__(self):
        import ctypes as C
        import pdwt_amd
        L = pdwt_amd.hip()
        for k, v in self.kw.items():
            cur = C.c_int()
            assert L.pdwt_debug_get(k.encode(), C.byref(cur)) == 0, k
            self.old[k] = cur.value
            assert L.pdwt_debug_set(k.encode(), int(v)) == 0, k
        return self

    def __exit__(self, *exc):
        import pdwt_amd
        L = pdwt_amd.hip()
        for k, v in self.old.items():
            L.pdwt_debug_set(k.encode(), v)
        return False


KIND = {"dwt2": dict(do_swt=0, ndim=2), "dwt1": dict(do_swt=0, ndim=1), "swt2": dict(do_swt=1, ndim=2), "swt1": dict(do_swt=1, ndim=1)}

GOLDEN_CASES = [
    "lena64_haar_L1", "u64_db4_L3_f32", "u64_db4_L3_f64", "odd63x65_db2_L2", "odd31x40_db3_L1",
    "odd50x37_sym4_L2", "odd45x77_bior2.4_L2", "r96x160_coif2_L3", "swt112_db7_L3", "swt64_db3_L3", "swt48x80_db2_L2",
    "swt64_haar_L2", "swt1d_6x128_sym4_L3", "b1d_5x256_sym8_L4", "b1d_3x77_db3_L2_odd", "b1d_1x200_db5_L3",
    "haar64_L3", "haar37x51_L2_odd", "haar1d_3x77_L3_odd", "haar1d_4x64_L2_f32",
    # one long bank per family, 2-D, several levels (F8 of tests/golden/make_golden.py)
    "long160x192_sym20_L2", "long128x160_coif5_L2", "long144x160_bior6.8_L3", "long96x144_rbio6.8_L2", "long120x168_db14_L2", "long112x80_bior3.9_L2",
    "swt96_sym8_L2",
]
