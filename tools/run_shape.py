#!/usr/bin/env python3
"""Run fwd+inv of one shape a few times (for rocprofv3 kernel traces). usage: run_shape.py Nr Nc wname levels [reps] [float32|float64] [swt]"""
import sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime in the process)
import pdwt_amd

nr, nc, wname, lev = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dt = sys.argv[6] if len(sys.argv) > 6 else "float32"
swt = 1 if len(sys.argv) > 7 else 0
x = np.random.RandomState(0).uniform(0, 255, (nr, nc)).astype(dt)
W = pdwt_amd.Wavelets(x, wname, lev, do_swt=swt)
for _ in range(reps):
    W.forward()
    W.inverse()
pdwt_amd.hip().pdwt_sync()
