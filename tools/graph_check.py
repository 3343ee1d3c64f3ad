#!/usr/bin/env python3
"""Digest of forward/inverse results over a set of configurations; run once with PDWT_GRAPH=0 and once with
PDWT_GRAPH=1 -- the two digests must be identical (tests/test_gpu_parity.py::test_graph_replay_is_bit_identical).
usage: PYTHONPATH=. PDWT_GRAPH=1 python tools/graph_check.py"""
import hashlib

import numpy as np
import pdwt_amd

CASES = [
    # shape, wname, levels, kwargs
    ((512, 512), "db4", 3, {}),
    ((256, 384), "sym8", 2, {}),
    ((300, 211), "db3", 2, {}),
    ((512, 512), "haar", 4, {}),
    ((256, 256), "db4", 3, dict(do_swt=1)),
    ((64, 4096), "db4", 4, dict(ndim=1)),
    ((128, 128), "db2", 2, dict(do_separable=0)),
    ((256, 256), "db7", 2, dict(do_cycle_spinning=1)),
]
h = hashlib.sha256()
rng = np.random.default_rng(7)
for dt in (np.float32, np.float64):
    for shape, wname, lev, kw in CASES:
        W = pdwt_amd.Wavelets(rng.standard_normal(shape).astype(dt), wname, lev, **kw)
        for rep in range(3):  # rep 0 records, reps 1-2 replay on fresh images
            img = rng.standard_normal(shape).astype(dt)
            W.set_image(img)
            W.forward()
            if not kw.get("do_cycle_spinning"):  # the shift is rand(): results differ run to run by design
                for c in W.coeffs:
                    h.update(np.ascontiguousarray(c).tobytes())
            W.soft_threshold(0.05)
            W.inverse()
            out = W.get_image()
            if not kw.get("do_cycle_spinning"):
                h.update(np.ascontiguousarray(out).tobytes())
            else:
                assert np.isfinite(out).all()
        if rep == 2 and wname == "db4" and not kw:  # new taps must drop the recorded launches
            f = np.array([0.5, 0.5, 0.25, -0.25, 0.125, 0.125, 0.0, 0.0], dtype=dt)
            W.set_filters_forward("custom8", f, f[::-1].copy())
            W.set_image(img)
            W.forward()
            for c in W.coeffs:
                h.update(np.ascontiguousarray(c).tobytes())
        W.close()
print("DIGEST", h.hexdigest())
