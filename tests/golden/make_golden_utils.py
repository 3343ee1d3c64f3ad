#!/opt/conda/bin/python3.9
"""Golden vectors for the coefficient utilities and custom filters, PyWavelets-based.

Run in the BUILD container only:   /opt/conda/bin/python3.9 tests/golden/make_golden_utils.py
Same conventions as make_golden.py (band order = PDWT's d_coeffs order, mode 'periodization').
 * hard / soft threshold: pywt.threshold(b, beta, 'hard'/'soft') on the detail bands (the random input has no
   |c| == beta tie, where pywt keeps and the reference zeroes)
 * proj_linf: clip to [-beta, beta]; shrink: b/(1+beta); group soft threshold: b*max(1-beta/||(h,v,d)||,0)
   per position and level -- closed forms of the reference kernels (src/common.cu:96-198, 346-371), evaluated
   here in float64 on the pywt coefficients
 * norm2sq: sum of squares of all bands
 * custom filters: the bior3.3 bank handed over as four explicit tap arrays; expected = pywt with a
   pywt.Wavelet built from the same arrays
 * cycle spinning: coefficients of the image circularly shifted by (sr, sc) = pywt on np.roll(x, (sr, sc))
"""
import os
import warnings

warnings.filterwarnings("ignore")
import numpy as np
import pywt

HERE = os.path.dirname(os.path.abspath(__file__))
MODE = "periodization"


def bands_of(c, L):
    out = [c[0]]
    for i in range(L):
        out += list(c[L - i])
    return out


def pack(prefix, bands):
    return {"%s%d" % (prefix, i): np.asarray(b, dtype=np.float64) for i, b in enumerate(bands)}


x = np.random.RandomState(21).uniform(0, 255, (64, 96))
w, L, beta = "db4", 3, 40.0
c = pywt.wavedec2(x, w, MODE, L)
b = bands_of(c, L)
out = dict(input=x, wname=w, levels=L, beta=beta, nbands=len(b))
out.update(pack("band", b))
det = lambda f: [b[0]] + [f(v) for v in b[1:]]
out.update(pack("hard", det(lambda v: pywt.threshold(v, beta, "hard"))))
out.update(pack("soft", det(lambda v: pywt.threshold(v, beta, "soft"))))
out.update(pack("proj", [np.clip(v, -beta, beta) for v in b]))           # do_thresh_appcoeffs = 1 (the default)
out.update(pack("shrink", [v / (1.0 + beta) for v in b]))                # do_thresh_appcoeffs = 1 (the default)
g = [b[0]]
for i in range(L):
    h, v, d = b[3 * i + 1: 3 * i + 4]
    n = np.sqrt(h * h + v * v + d * d)
    r = np.where(n == 0, 0.0, np.maximum(1 - beta / np.where(n == 0, 1, n), 0.0))
    g += [h * r, v * r, d * r]
out.update(pack("group", g))
out["norm2sq"] = sum((v * v).sum() for v in b)
out["norm1"] = sum(np.abs(v).sum() for v in b)
# normalize=1: beta/sqrt(2)^(level) on the details of level `level` (1 = finest)
sn = [b[0]]
for i in range(L):
    bl = beta / np.sqrt(2.0) ** (i + 1)
    sn += [pywt.threshold(v, bl, "soft") for v in b[3 * i + 1: 3 * i + 4]]
out.update(pack("softnorm", sn))
np.savez_compressed(os.path.join(HERE, "utils64x96_db4_L3.npz"), **out)

# custom filters
bw = pywt.Wavelet("bior3.3")
cw = pywt.Wavelet("custom_bior33", filter_bank=[bw.dec_lo, bw.dec_hi, bw.rec_lo, bw.rec_hi])
x = np.random.RandomState(22).randn(80, 64)
c = pywt.wavedec2(x, cw, MODE, 2)
b = bands_of(c, 2)
np.savez_compressed(os.path.join(HERE, "custom80x64_bior33_L2.npz"), input=x, levels=2, nbands=len(b), dec_lo=np.array(bw.dec_lo), dec_hi=np.array(bw.dec_hi),
                    rec_lo=np.array(bw.rec_lo), rec_hi=np.array(bw.rec_hi), recon=pywt.waverec2(c, cw, MODE), **pack("band", b))

# cycle spinning: a known shift
x = np.random.RandomState(23).randn(48, 72)
sr, sc = 5, 61
c = pywt.wavedec2(np.roll(x, (sr, sc), axis=(0, 1)), "db3", MODE, 2)
b = bands_of(c, 2)
np.savez_compressed(os.path.join(HERE, "shift48x72_db3_L2.npz"), input=x, wname="db3", levels=2, sr=sr, sc=sc, nbands=len(b), shifted=np.roll(x, (sr, sc), axis=(0, 1)),
                    **pack("band", b))
print("ok")
