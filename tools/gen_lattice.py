#!/usr/bin/env python3
"""Generate pdwt_amd/csrc/lattice_table.inc: the paraunitary LATTICE factorisation of the orthogonal banks of filters_table.inc.

Why: a 2-channel orthogonal analysis bank of 2K taps, in polyphase form on sample pairs, factors into K plane rotations separated by
one-sample delays (Vaidyanathan, "Multirate Systems and Filter Banks", ch. 6.4):

    E(z) = R_{K-1} L(z) R_{K-2} ... L(z) R_1 L(z) M0,      L(z) = diag(1, z^-1),   R_i ~ [[1, -t_i], [t_i, 1]]

which evaluates one (lo, hi) output pair in 2K + 2 multiply-adds instead of the 4K of the direct form -- the lever the
double-precision level kernels of dwt_lat.hip use for the COLUMN pass of long banks (BASELINE config 5: db20, K = 20: 42 instead of
80).  The synthesis bank is the transposed cascade.

Numerics.  The downward recursion that peels the rotations off the taps is ill-conditioned (db20: ratios of taps of 1e-13), and the
table's double-precision taps are orthonormal only to ~1e-17.  So the tool (i) moves the taps, in 120-digit arithmetic, to the NEAREST
exactly orthonormal low-pass filter (minimum-norm Newton steps on sum_k h[k] h[k+2m] = delta_m; the move is reported: <= 3e-17 for the
db / coif banks), (ii) runs the recursion on that filter in 120 digits, (iii) rounds M0, the inverse's end matrix and the t_i to double
and (iv) checks the double-precision lattice against the double-precision direct form on a random periodic signal.  A bank is emitted
only when the move is <= 1e-15 and the check <= 1e-13 (pywt's sym banks are orthonormal to ~1e-12 only: their nearest orthonormal
filter is 5e-12 away, which would show at the 1e-10 parity tolerance -- they stay on the direct-form kernels).

Every stage is a scaled rotation, so the evaluation is backward stable whatever the size of t_i (db20: 8e-6 ... 2.6e6).

Usage:  python3 tools/gen_lattice.py          (needs mpmath and numpy; reads pdwt_amd/csrc/filters_table.inc)
"""
import hashlib
import os
import re

import mpmath as mp
import numpy as np

mp.mp.dps = 120
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_banks():
    txt = open(os.path.join(ROOT, "pdwt_amd", "csrc", "filters_table.inc")).read()
    banks = []
    for m in re.finditer(r'PDWT_FILTER\("([^"]+)",\s*(\d+),\s*\{([^}]*)\},\s*\{([^}]*)\},\s*\{([^}]*)\},\s*\{([^}]*)\}\)', txt):
        name, hlen = m.group(1), int(m.group(2))
        arrs = [[float(v) for v in m.group(i).split(",")] for i in (3, 4, 5, 6)]
        assert all(len(a) == hlen for a in arrs), name
        banks.append((name, hlen, arrs))
    return banks


def refine(dl):
    N, K = len(dl), len(dl) // 2
    h = mp.matrix([mp.mpf(v) for v in dl])
    nrm = None
    for _ in range(8):
        g, J = mp.matrix(K, 1), mp.matrix(K, N)
        for m in range(K):
            s = mp.mpf(0)
            for k in range(N - 2 * m):
                s += h[k] * h[k + 2 * m]
                J[m, k] += h[k + 2 * m]
                J[m, k + 2 * m] += h[k]
            g[m] = s - (1 if m == 0 else 0)
        nrm = max(abs(v) for v in g)
        if nrm < mp.mpf(10) ** (-100):
            break
        h = h - J.T * mp.lu_solve(J * J.T, g)
    return h, nrm


def lattice(h, sgn):
    N, K = len(h), len(h) // 2
    g = [sgn * (-1) ** k * h[N - 1 - k] for k in range(N)]
    E = [[[h[2 * k + 1], h[2 * k]], [g[2 * k + 1], g[2 * k]]] for k in range(K)]
    ts, resid = [], mp.mpf(0)
    for m in range(K - 1, 0, -1):
        (a0, b0), (c0, d0) = E[0]
        t = c0 / a0 if abs(a0) >= abs(b0) else d0 / b0
        ts.append(t)
        R0 = [[E[k][0][0] + t * E[k][1][0], E[k][0][1] + t * E[k][1][1]] for k in range(m + 1)]
        R1 = [[-t * E[k][0][0] + E[k][1][0], -t * E[k][0][1] + E[k][1][1]] for k in range(m + 1)]
        resid = max(resid, abs(R1[0][0]), abs(R1[0][1]), abs(R0[m][0]), abs(R0[m][1]))
        E = [[R0[k], R1[k + 1]] for k in range(m)]
    ts = ts[::-1]
    S2 = mp.mpf(1)
    for t in ts:
        S2 *= 1 + t * t
    M0 = mp.matrix([[E[0][i][j] / S2 for j in range(2)] for i in range(2)])
    Mi = M0 ** -1 / S2
    return K, M0, Mi, ts, resid


def lat_fwd(x, M0, ts):
    K = len(ts) + 1
    sig, D = (K - 1) & 1, (K - 1) >> 1
    xs = np.roll(x, -sig)
    e, o = xs[0::2].copy(), xs[1::2].copy()
    u = M0[0] * e + M0[1] * o
    v = M0[2] * e + M0[3] * o
    for t in ts:
        vd = np.roll(v, 1)
        u, v = u - t * vd, t * u + vd
    return np.roll(u, -D), np.roll(v, -D)


def lat_inv(lo, hi, Mi, ts):
    K = len(ts) + 1
    sig, D = (K - 1) & 1, (K - 1) >> 1
    a, b = np.roll(lo, D).copy(), np.roll(hi, D).copy()
    for t in ts[::-1]:
        a, b = a + t * b, b - t * a
        a = np.roll(a, 1)
    e = np.roll(Mi[0] * a + Mi[1] * b, -(K - 1))
    o = np.roll(Mi[2] * a + Mi[3] * b, -(K - 1))
    x = np.empty(2 * len(lo))
    x[0::2], x[1::2] = e, o
    return np.roll(x, sig)


def direct_fwd(x, dl, dh):
    N, n = len(dl), len(x)
    C = N // 2 - 1
    idx = (2 * np.arange(n // 2)[:, None] - C + np.arange(N)[None, :]) % n
    w = x[idx]
    return w @ np.array(dl)[::-1], w @ np.array(dh)[::-1]


def main():
    out = os.path.join(ROOT, "pdwt_amd", "csrc", "lattice_table.inc")
    rs = np.random.RandomState(5)
    x = rs.standard_normal(1024)
    lines, log = [], []
    for name, hlen, (dl, dh, rl, rh) in read_banks():
        if hlen < 4 or hlen % 2 or not (name.startswith("db") or name.startswith("sym") or name.startswith("coif")):
            continue
        sgn = None
        for s in (1, -1):
            if max(abs(s * (-1) ** k * dl[hlen - 1 - k] - dh[k]) for k in range(hlen)) == 0.0:
                sgn = s
        rec_ok = all(rl[k] == dl[hlen - 1 - k] and rh[k] == dh[hlen - 1 - k] for k in range(hlen))
        if sgn is None or not rec_ok:
            log.append("%-7s skipped: dec_hi is not the exact mirror of dec_lo / rec is not the exact reverse of dec" % name)
            continue
        h, nrm = refine(dl)
        move = float(max(abs(h[k] - mp.mpf(dl[k])) for k in range(hlen)))
        K, M0, Mi, ts, resid = lattice(h, sgn)
        m0 = [float(M0[i, j]) for i in range(2) for j in range(2)]
        mi = [float(Mi[i, j]) for i in range(2) for j in range(2)]
        tf = [float(t) for t in ts]
        lo, hi = lat_fwd(x, m0, tf)
        rlo, rhi = direct_fwd(x, dl, dh)
        err_f = max(np.abs(lo - rlo).max(), np.abs(hi - rhi).max())
        err_i = np.abs(lat_inv(rlo, rhi, mi, tf) - x).max()
        ok = move <= 1e-15 and err_f <= 1e-13 and err_i <= 1e-13
        log.append("%-7s K=%2d  move %.1e  recursion residual %.0e  max|t| %.2e  lattice vs direct %.1e  inverse %.1e  %s"
                   % (name, K, move, float(resid), max(abs(t) for t in tf), err_f, err_i, "emitted" if ok else "NOT emitted"))
        if ok:
            fmt = lambda a: ", ".join(repr(v) for v in a)
            lines.append('PDWT_LATTICE("%s", %d,\n  {%s},\n  {%s},\n  {%s})' % (name, K, fmt(m0), fmt(mi), fmt(tf)))
    body = "\n".join(lines) + "\n"
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_lattice.py from filters_table.inc (sha256 %s) -- do not edit.\n"
                % hashlib.sha256(open(os.path.join(ROOT, "pdwt_amd", "csrc", "filters_table.inc"), "rb").read()).hexdigest()[:16])
        f.write("// One entry per orthogonal bank whose lattice reproduces the direct form to <= 1e-13 in double precision:\n")
        f.write("// PDWT_LATTICE(name, K, M0[4] (analysis: first stage, row-major, all scalings folded in), MI[4] (synthesis: last stage), t[K-1])\n")
        f.write("// analysis, pairs (e, o) = (x[2n+s], x[2n+s+1]), s = (K-1)&1:   (u, v) = M0 (e, o);  for i = 1..K-1: vd = delay(v); (u, v) = (u - t_i vd, t_i u + vd);\n")
        f.write("//   out_lo[n - ((K-1)>>1)] = u, out_hi[...] = v.   synthesis: (a, b) = (lo, hi); for i = K-1..1: (a, b) = (a + t_i b, b - t_i a); a = delay(a);  (e, o) = MI (a, b)\n")
        f.write(body)
    print("\n".join(log))
    print("wrote %s: %d banks" % (out, len(lines)))


if __name__ == "__main__":
    main()
