#!/usr/bin/env python3
"""gp.hessian(testing, is_gpu=True, precision=...) from host numpy arrays to a host (M, D, D) float64 array:
BASELINE config 5's shape (N=300, D=16, 1e6 rows: 2 GB of float64 results) in both compute precisions.

    python tools/hessian_host_timing.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_emulator_amd import GaussianProcess, _lib  # noqa: E402
from bench import synthetic_inputs  # noqa: E402

_lib.bind_near_device(0)
N, D, M = 300, 16, 1000000
inputs, testing, theta, invQ, invQt = synthetic_inputs(1, N, D, M)
gp = GaussianProcess(inputs, [])
gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
ref = gp.hessian(testing[:64])
for prec in (np.float64, np.float32):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        h = gp.hessian(testing, is_gpu=True, precision=prec)
        ts.append(time.perf_counter() - t0)
        e = float(np.max(np.abs(h[:64] - ref)) / np.max(np.abs(ref)))
        assert h.dtype == np.float64
        del h
    print("hessian host to host, compute %s: %s ms -> best %.1f ms = %.2e rows/s; parity %.1e" % (
        np.dtype(prec).name, ", ".join("%.1f" % (t * 1e3) for t in ts), min(ts) * 1e3, M / min(ts), e))
