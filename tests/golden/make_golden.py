#!/opt/conda/bin/python3.9
"""Generate the golden vectors under tests/golden/ with PyWavelets (mode='periodization').

Run in the BUILD container only:   /opt/conda/bin/python3.9 tests/golden/make_golden.py
(PyWavelets 1.1.1 lives in the conda python there; neither it nor /root/reference exist on the
GPU box, so the outputs -- data only -- are committed.)

Why PyWavelets: the reference has no tests or expected outputs of its own;
its contract is "results compatible with ... Python pywt" with periodic extension
(reference README.md:25,28) and BASELINE.json names PyWavelets as the parity oracle.

Band order stored = PDWT's d_coeffs order (SURVEY.md 8c "pywt <-> PDWT mapping"):
  2D : band0 = cA_L ; band(3i+1..3i+3) = (cH, cV, cD) of level i+1 (finest = 1)
  1D : band0 = cA_L ; band(i+1) = cD of level i+1
All expected arrays are float64 computed from the float64 view of the stored input.
`lena.dat` is the reference's own input fixture (test/lena.dat, raw 512x512 float32), copied as data.
"""
import hashlib
import os
import shutil
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np
import pywt

HERE = os.path.dirname(os.path.abspath(__file__))
MODE = "periodization"


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def dwt2_bands(x, w, L):
    c = pywt.wavedec2(x, w, MODE, L)
    bands = [c[0]]
    for i in range(L):  # level i+1 <-> c[L-i]
        bands += list(c[L - i])
    return bands, pywt.waverec2(c, w, MODE)


def dwt1_bands(x, w, L):
    c = pywt.wavedec(x, w, MODE, L, axis=-1)
    bands = [c[0]] + [c[L - i] for i in range(L)]
    return bands, pywt.waverec(c, w, MODE, axis=-1)


def swt2_bands(x, w, L):
    c = pywt.swt2(x, w, L)  # coarsest first
    bands = [c[0][0]]
    for i in range(L):
        bands += list(c[L - 1 - i][1])
    return bands, pywt.iswt2(c, w)


def swt1_bands(x, w, L):
    c = pywt.swt(x, w, L, axis=-1)  # [(cA_L,cD_L),...,(cA_1,cD_1)]
    bands = [c[0][0]] + [c[L - 1 - i][1] for i in range(L)]
    rec = np.stack([pywt.iswt([(a[r], d[r]) for a, d in c], w) for r in range(x.shape[0])])
    return bands, rec


def pack(bands, dt=np.float64):
    return {"band%d" % i: np.asarray(b, dtype=dt) for i, b in enumerate(bands)}


def case(name, x, w, L, kind, in_dtype):
    xs = x.astype(in_dtype)           # what the library is fed
    x64 = xs.astype(np.float64)       # what pywt is fed
    fn = {"dwt2": dwt2_bands, "dwt1": dwt1_bands, "swt2": swt2_bands, "swt1": swt1_bands}[kind]
    bands, rec = fn(x64, w, L)
    # expected values of float32-input cases are stored as float32 (6e-8 relative, far below the
    # 1e-5 bar) to keep the fixtures small
    save(name, input=xs, wname=w, levels=L, kind=kind, nbands=len(bands), recon=np.asarray(rec, in_dtype)[: x.shape[0], : x.shape[1]], **pack(bands, in_dtype))


rs = np.random.RandomState

# F1: the reference's own image fixture, haar L1 (config C1) -- 64x64 crop stored in full,
# full 512x512 as per-band float64 sums + a strided sample.
src = "/root/reference/test/lena.dat"
dst = os.path.join(HERE, "lena.dat")
if os.path.exists(src):
    shutil.copyfile(src, dst)
lena = np.fromfile(dst, dtype=np.float32).reshape(512, 512)
assert hashlib.sha256(lena.tobytes()).hexdigest().startswith("3ef6d848")
case("lena64_haar_L1", lena[:64, :64], "haar", 1, "dwt2", np.float32)
bands, rec = dwt2_bands(lena.astype(np.float64), "haar", 1)
save("lena512_haar_L1_summary", wname="haar", levels=1,
     sums=np.array([b.sum() for b in bands]), abssums=np.array([np.abs(b).sum() for b in bands]),
     sample=np.stack([b[::16, ::16] for b in bands]), recon_maxerr=np.abs(rec - lena).max())
bands, rec = dwt2_bands(lena.astype(np.float64), "db4", 3)
save("lena512_db4_L3_summary", wname="db4", levels=3,
     sums=np.array([b.sum() for b in bands]), abssums=np.array([np.abs(b).sum() for b in bands]),
     norm1=sum(np.abs(b).sum() for b in bands), recon_maxerr=np.abs(rec - lena).max(),
     **{"sample%d" % i: b[::8, ::8] for i, b in enumerate(bands)})

# F2: 64x64 uniform[0,255) db4 L3, f32 and f64 inputs
x = rs(0).uniform(0, 255, (64, 64))
case("u64_db4_L3_f32", x, "db4", 3, "dwt2", np.float32)
case("u64_db4_L3_f64", x, "db4", 3, "dwt2", np.float64)
# F3: odd sizes, odd / even half filter length
case("odd63x65_db2_L2", rs(3).randn(63, 65), "db2", 2, "dwt2", np.float64)
case("odd31x40_db3_L1", rs(4).randn(31, 40), "db3", 1, "dwt2", np.float64)
case("odd50x37_sym4_L2", rs(5).randn(50, 37), "sym4", 2, "dwt2", np.float32)
case("odd45x77_bior2.4_L2", rs(6).randn(45, 77), "bior2.4", 2, "dwt2", np.float64)
case("r96x160_coif2_L3", rs(7).randn(96, 160), "coif2", 3, "dwt2", np.float32)
# F4: SWT
case("swt112_db7_L3", rs(8).uniform(0, 255, (112, 112)), "db7", 3, "swt2", np.float32)
case("swt64_db3_L3", rs(19).randn(64, 64), "db3", 3, "swt2", np.float64)
case("swt48x80_db2_L2", rs(9).randn(48, 80), "db2", 2, "swt2", np.float64)
case("swt64_haar_L2", rs(10).randn(64, 64), "haar", 2, "swt2", np.float64)
case("swt1d_6x128_sym4_L3", rs(11).randn(6, 128), "sym4", 3, "swt1", np.float64)
# F5: batched 1D
case("b1d_5x256_sym8_L4", rs(1).randn(5, 256), "sym8", 4, "dwt1", np.float32)
case("b1d_3x77_db3_L2_odd", rs(12).randn(3, 77), "db3", 2, "dwt1", np.float64)
case("b1d_1x200_db5_L3", rs(13).randn(1, 200), "db5", 3, "dwt1", np.float64)
# Haar multi-level (2D even, 2D odd, 1D odd)
case("haar64_L3", rs(14).randn(64, 64), "haar", 3, "dwt2", np.float64)
case("haar37x51_L2_odd", rs(15).randn(37, 51), "haar", 2, "dwt2", np.float64)
case("haar1d_3x77_L3_odd", rs(16).randn(3, 77), "haar", 3, "dwt1", np.float64)
case("haar1d_4x64_L2_f32", rs(17).uniform(0, 255, (4, 64)), "haar", 2, "dwt1", np.float32)

# F6: long taps f64 pipeline: forward -> soft_threshold(0.5) on details -> norm1 -> inverse
x = rs(2).randn(160, 192)
w, L, beta = "db20", 2, 0.5
c = pywt.wavedec2(x, w, MODE, L)
bands = [c[0]] + [b for i in range(L) for b in c[L - i]]
n1_before = sum(np.abs(b).sum() for b in bands)
ct = [c[0]] + [tuple(pywt.threshold(b, beta, "soft") for b in lev) for lev in c[1:]]
tb = [ct[0]] + [b for i in range(L) for b in ct[L - i]]
n1_after = sum(np.abs(b).sum() for b in tb)
save("pipe160x192_db20_L2_f64", input=x, wname=w, levels=L, kind="dwt2", nbands=len(bands), beta=beta,
     norm1_before=n1_before, norm1_after=n1_after, recon=pywt.waverec2(c, w, MODE),
     recon_thresh=pywt.waverec2(ct, w, MODE), **pack(bands))

# F8: one LONG bank per family in 2-D over several levels (the 72-wavelet fixture F7 pins the tap table through a level-1
# 1-D transform only; these pin the long banks -- and the multi-level 2-D index arithmetic on them -- against pywt directly)
case("long160x192_sym20_L2", rs(21).randn(160, 192), "sym20", 2, "dwt2", np.float32)
case("long128x160_coif5_L2", rs(22).randn(128, 160), "coif5", 2, "dwt2", np.float64)
case("long144x160_bior6.8_L3", rs(23).randn(144, 160), "bior6.8", 3, "dwt2", np.float64)
case("long96x144_rbio6.8_L2", rs(24).randn(96, 144), "rbio6.8", 2, "dwt2", np.float32)
case("long120x168_db14_L2", rs(25).randn(120, 168), "db14", 2, "dwt2", np.float64)
case("long112x80_bior3.9_L2", rs(26).randn(112, 80), "bior3.9", 2, "dwt2", np.float64)
case("swt96_sym8_L2", rs(27).randn(96, 96), "sym8", 2, "swt2", np.float32)

# F7: every one of the 72 wavelets, one short 1D signal each (pins the filter table + index math
# for every filter length 2..40) -- level 1 forward + the round trip.
from importlib import import_module
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools"))
names = import_module("gen_filters").NAMES
x = rs(18).randn(2, 256)
allb = {}
for n in names:
    cA, cD = pywt.dwt(x, n, MODE, axis=-1)
    allb["A_" + n] = cA
    allb["D_" + n] = cD
save("all72_1d_2x256_L1", input=x, names=np.array(names), **allb)
print("done")
