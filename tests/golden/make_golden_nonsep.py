#!/usr/bin/env python3
"""Golden vectors for the NON-separable transform with four arbitrary hlen x hlen kernels.

PyWavelets has no non-separable transform, so these vectors are an INDEPENDENT float64 evaluation of the reference's defining sums
(src/nonseparable.cu:114-170 forward, :176-226 inverse, :304-352 a-trous forward, :358-401 a-trous inverse), written array-wise --
whole-image circular shifts and gathers -- not sample by sample like the oracle's C restatement (oracle/pdwt_oracle_impl.h) or the HIP
kernel (pdwt_amd/csrc/nonsep.hip).  Both are checked against these files (tests/test_oracle_utils.py, tests/test_gpu_parity.py).
numpy only:   python tests/golden/make_golden_nonsep.py

Definitions restated from the reference (K = one of the four kernels (LL, LH, HL, HH), row-major [y][x]; bands A, H, V, D in that order):
  forward, decimated:  c = hlen/2 - 1 (even hlen) | hlen/2 (odd);  xe = x with its last row / column repeated once when that size is odd,
                       then periodic;   out[gy, gx] = sum_{jy, jx < hlen} xe[2 gy - c + jy, 2 gx - c + jx] * K[hlen-1-jy, hlen-1-jx]
  inverse, decimated:  h2 = hlen/2;  c = h2/2;  g = o + 1 when h2 is even, o otherwise;  p = g/2, off = 1 - (g & 1);  taps j = 0 .. h2-1 when
                       h2 is odd, j = 0 .. 2c-1 when it is even;
                       out[oy, ox] = sum over the four bands of sum_{jy, jx} band[(py - c + jy) mod Nr, (px - c + jx) mod Nc]
                                                                             * K[hlen-1-(2 jy + offy), hlen-1-(2 jx + offx)]
  forward, a-trous, level l:  f = 2^(l-1);  c = (hlen/2 - 1 | hlen/2) * f;
                       out[gy, gx] = sum_{jy, jx < hlen} x[(gy - c + f jy) mod Nr, (gx - c + f jx) mod Nc] * K[hlen-1-jy, hlen-1-jx]
  inverse, a-trous:    c = (hlen/2) * f;  out = sum over the four bands of the same sum over hlen x hlen taps, each product divided by 4
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def div2(n):
    return (n + 1) // 2


def fwd_level(x, K):
    """-> [A, H, V, D] of one decimated level"""
    hlen = K[0].shape[0]
    c = hlen // 2 if hlen & 1 else hlen // 2 - 1
    nr, nc = x.shape
    xe = x
    if nr & 1:
        xe = np.concatenate([xe, xe[-1:, :]], axis=0)
    if nc & 1:
        xe = np.concatenate([xe, xe[:, -1:]], axis=1)
    out = [np.zeros((div2(nr), div2(nc))) for _ in range(4)]
    for jy in range(hlen):
        for jx in range(hlen):
            s = np.roll(xe, (c - jy, c - jx), axis=(0, 1))[::2, ::2]  # s[gy, gx] = xe[2 gy - c + jy, 2 gx - c + jx]
            for b in range(4):
                out[b] = out[b] + s * K[b][hlen - 1 - jy, hlen - 1 - jx]
    return out


def inv_level(bands, K, nro, nco):
    """four bands (Nr x Nc) -> image (nro x nco)"""
    hlen = K[0].shape[0]
    h2 = hlen // 2
    c = h2 // 2
    ntap = h2 if (h2 & 1) else 2 * c        # hL + hR + 1 (src/nonseparable.cu:183-193)
    shift = 0 if (h2 & 1) else 1
    nr, nc = bands[0].shape
    gy, gx = np.arange(nro) + shift, np.arange(nco) + shift
    py, px, offy, offx = gy // 2, gx // 2, 1 - (gy & 1), 1 - (gx & 1)
    out = np.zeros((nro, nco))
    for jy in range(ntap):
        iy = (py - c + jy) % nr
        ky = hlen - 1 - (2 * jy + offy)
        for jx in range(ntap):
            ix = (px - c + jx) % nc
            kx = hlen - 1 - (2 * jx + offx)
            for b in range(4):
                out = out + bands[b][np.ix_(iy, ix)] * K[b][np.ix_(ky, kx)]
    return out


def swt_fwd_level(x, K, level):
    hlen = K[0].shape[0]
    f = 1 << (level - 1)
    c = (hlen // 2 if hlen & 1 else hlen // 2 - 1) * f
    out = [np.zeros_like(x) for _ in range(4)]
    for jy in range(hlen):
        for jx in range(hlen):
            s = np.roll(x, (c - f * jy, c - f * jx), axis=(0, 1))
            for b in range(4):
                out[b] = out[b] + s * K[b][hlen - 1 - jy, hlen - 1 - jx]
    return out


def swt_inv_level(bands, K, level):
    hlen = K[0].shape[0]
    f = 1 << (level - 1)
    c = (hlen // 2) * f
    out = np.zeros_like(bands[0])
    for jy in range(hlen):
        for jx in range(hlen):
            for b in range(4):
                out = out + np.roll(bands[b], (c - f * jy, c - f * jx), axis=(0, 1)) * K[b][hlen - 1 - jy, hlen - 1 - jx] / 4
    return out


def forward(x, K, levels, swt):
    a, det = x, []
    for lev in range(levels):
        A, H, V, D = swt_fwd_level(a, K, lev + 1) if swt else fwd_level(a, K)
        det += [H, V, D]
        a = A
    return [a] + det


def inverse(bands, K, levels, swt, shape):
    sizes = [shape]
    for _ in range(levels):
        sizes.append(sizes[-1] if swt else (div2(sizes[-1][0]), div2(sizes[-1][1])))
    a = bands[0]
    for i in range(levels - 1, -1, -1):
        four = [a, bands[3 * i + 1], bands[3 * i + 2], bands[3 * i + 3]]
        a = swt_inv_level(four, K, i + 1) if swt else inv_level(four, K, sizes[i][0], sizes[i][1])
    return a


CASES = [  # name, hlen, shape, levels, swt
    ("nsep_dec_h6_48x64_L2", 6, (48, 64), 2, 0),
    ("nsep_dec_h4_33x47_L2_odd", 4, (33, 47), 2, 0),   # odd sizes at both levels (33 -> 17 -> 9, 47 -> 24 -> 12)
    ("nsep_dec_h5_40x56_L1", 5, (40, 56), 1, 0),       # odd kernel size (custom filters only)
    ("nsep_swt_h4_32x48_L2", 4, (32, 48), 2, 1),
    ("nsep_swt_h5_40x56_L2", 5, (40, 56), 2, 1),
]

if __name__ == "__main__":
    rs = np.random.RandomState(88)
    for name, hlen, shape, levels, swt in CASES:
        kf = [rs.randn(hlen, hlen) for _ in range(4)]
        ki = [rs.randn(hlen, hlen) for _ in range(4)]
        x = rs.randn(*shape)
        b = forward(x, kf, levels, swt)
        rec = inverse(b, ki, levels, swt, shape)  # (arbitrary kernels: NOT a reconstruction of x, just the synthesis operator's output)
        d = dict(input=x, levels=levels, swt=swt, hlen=hlen, nbands=len(b), recon=rec)
        for q in range(4):
            d["kf%d" % q] = kf[q]
            d["ki%d" % q] = ki[q]
        for q, v in enumerate(b):
            d["band%d" % q] = v
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, [v.shape for v in b[:2]], float(np.abs(rec).max()))
    print("ok")
