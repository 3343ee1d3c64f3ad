#!/usr/bin/env python3
"""Timeline of the cascade kernels from their own clocks (diagnostic build: tools/build_trace.sh).

  PDWT_LIBDIR=$PWD/pdwt_amd/lib_trace python tools/casc_trace.py [--size 4096] [--wname db4] [--levels 3] [--md out.md]

Every wave of k_fwd2d_casc / k_inv2d_cascw records the 100 MHz real-time counter at entry, after its prologue, after its first
loads landed, after the first (half) super-bodies, at loop exit and after its final drain (casc_dev.hpp: CASC_TRACE).  This tool
runs a few forward / inverse pairs, reads the records of the LAST launch of each kernel and prints, relative to the first
wave's entry: when waves start (launch ramp), how long each phase takes (median and 5/95 percentiles over waves) and when
waves end (tail).  Times in microseconds (10 ns resolution).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pdwt_amd  # noqa: E402
from pdwt_amd import _native as N  # noqa: E402

TRACE_OFF_FLOATS = 1 << 20
NAMES_F = ["entry", "ring prologue computed", "first body rows landed", "first half super-body", "first super-body", "loop exit", "drained"]
NAMES_I = ["entry", "L3 prologue done", "warm-up rows landed", "super-body 0", "super-body 1", "loop exit", "drained"]


def read_trace(W, nwaves, which):
    buf = np.zeros(nwaves * 8, dtype=np.uint64)
    tmp = W._L.pdwt_wavelets_tmp_int_ptr(W._h)
    assert N.hip().pdwt_memcpy_d2h(buf.ctypes.data_as(C.c_void_p), C.c_void_p(tmp + 4 * TRACE_OFF_FLOATS * (1 + which)), buf.nbytes) == 0
    return buf.reshape(nwaves, 8)


def clear_trace(W, nwaves):
    tmp = W._L.pdwt_wavelets_tmp_int_ptr(W._h)
    for which in (0, 1):
        assert N.hip().pdwt_memset(C.c_void_p(tmp + 4 * TRACE_OFF_FLOATS * (1 + which)), 0, nwaves * 64) == 0


def pct(a, q):
    return float(np.percentile(a, q))


def report(title, rec, names, out, W=None):
    ids = np.nonzero(rec[:, 0] != 0)[0]
    rec = rec[rec[:, 0] != 0]
    t = rec[:, :7].astype(np.int64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    out.append("### %s: %d waves recorded\n" % (title, len(rec)))
    out.append("| point | first wave | 5 %% | median | 95 %% | last wave |\n|---|---|---|---|---|---|")
    for k, n in enumerate(names):
        col = us[:, k]
        ok = t[:, k] != 0
        if not ok.any():
            continue
        col = col[ok]
        out.append("| %s | %.2f | %.2f | %.2f | %.2f | %.2f |" % (n, col.min(), pct(col, 5), pct(col, 50), pct(col, 95), col.max()))
    out.append("")
    out.append("| phase (per wave) | 5 % | median | 95 % |\n|---|---|---|---|")
    for k in range(1, 7):
        ok = (t[:, k] != 0) & (t[:, k - 1] != 0)
        if not ok.any():
            continue
        d = (t[ok, k] - t[ok, k - 1]) / 100.0
        out.append("| %s -> %s | %.2f | %.2f | %.2f |" % (names[k - 1], names[k], pct(d, 5), pct(d, 50), pct(d, 95)))
    life = us[:, 6] - us[:, 0]
    out.append("| whole wave | %.2f | %.2f | %.2f |" % (pct(life, 5), pct(life, 50), pct(life, 95)))
    out.append("")
    out.append("kernel span by these clocks: %.2f us (first entry -> last drain)\n" % us[:, 6].max())
    if W:
        # who is late: end time by wave position in the workgroup, by XCD (workgroup id mod 8), and the steps / rows each position runs
        kw, xcd = ids % W, (ids // W) % 8
        aux = rec[:, 7].astype(np.int64)
        out.append("| wave in workgroup | " + " | ".join(str(k) for k in range(W)) + " |\n|" + "---|" * (W + 1))
        out.append("| median end (us) | " + " | ".join("%.1f" % np.median(us[kw == k, 6]) for k in range(W)) + " |")
        out.append("| max end (us) | " + " | ".join("%.1f" % us[kw == k, 6].max() for k in range(W)) + " |")
        out.append("| median life (us) | " + " | ".join("%.1f" % np.median((us[:, 6] - us[:, 0])[kw == k]) for k in range(W)) + " |")
        out.append("| rows / steps (aux lo, hi; median) | " + " | ".join("%d/%d" % (np.median(aux[kw == k] & 0xffffffff), np.median(aux[kw == k] >> 32)) for k in range(W)) + " |")
        out.append("")
        out.append("| XCD | " + " | ".join(str(k) for k in range(8)) + " |\n|" + "---|" * 9)
        out.append("| median end (us) | " + " | ".join("%.1f" % np.median(us[xcd == k, 6]) for k in range(8)) + " |")
        out.append("| max end (us) | " + " | ".join("%.1f" % us[xcd == k, 6].max() for k in range(8)) + " |")
        out.append("")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--wname", default="db4")
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--md", default=None)
    ap.add_argument("--fw", type=int, default=16, help="waves per workgroup of the forward launch (for the per-position tables)")
    ap.add_argument("--iw", type=int, default=16, help="... of the inverse launch")
    ap.add_argument("--seq", default="b2b", choices=["b2b", "wr", "rd"])
    a = ap.parse_args()
    import torch
    x = torch.rand(a.size, a.size, device="cuda") * 255
    W = pdwt_amd.Wavelets(x, a.wname, a.levels)
    nwaves = 1 << 14
    out = ["# in-kernel timeline, %dx%d %s L%d (us relative to the first wave's entry; lib: %s)" % (a.size, a.size, a.wname, a.levels, N.LIBDIR) + " seq=" + a.seq]
    for _ in range(a.reps):
        W.forward()
        W.inverse()
    W.sync()
    # --seq: what runs right before the traced launch.  "b2b": the launches of a running forward / inverse alternation (what the
    # bench times); "wr": the same kernels after an idle device; "rd": after an idle device, the same direction twice in a row
    clear_trace(W, nwaves)
    if a.seq == "b2b":
        for _ in range(a.reps):
            W.forward()
            W.inverse()
        W.sync()
        f, i = read_trace(W, nwaves, 0), read_trace(W, nwaves, 1)
    else:
        if a.seq == "rd":
            W.forward()
        W.forward()
        W.sync()
        f = read_trace(W, nwaves, 0)
        if a.seq == "rd":
            W.inverse()
            W.state = pdwt_amd.wavelets.W_FORWARD  # the class refuses a second inverse(); the bands are still intact
        W.inverse()
        W.sync()
        i = read_trace(W, nwaves, 1)
    if not f[:, 0].any():
        sys.exit("no trace records: is PDWT_LIBDIR pointing at the -DPDWT_CASC_TRACE build?")
    report("forward cascade (k_fwd2d_casc)", f, NAMES_F, out, a.fw)
    report("inverse cascade (k_inv2d_cascw / k_inv2d_casc3)", i, NAMES_I, out, a.iw)
    if a.seq == "b2b":
        tf, ti = f[f[:, 0] != 0][:, :7].astype(np.int64), i[i[:, 0] != 0][:, :7].astype(np.int64)
        out.append("back to back: forward first entry -> inverse first entry %.2f us; forward last drain -> inverse first entry %.2f us; "
                   "forward first entry -> inverse last drain %.2f us\n"
                   % ((ti[:, 0].min() - tf[:, 0].min()) / 100.0, (ti[:, 0].min() - tf[:, 6].max()) / 100.0, (ti[:, 6].max() - tf[:, 0].min()) / 100.0))
    txt = "\n".join(out)
    print(txt)
    if a.md:
        open(a.md, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
