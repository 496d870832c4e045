#!/usr/bin/env python3
"""gp.predict(testing, is_gpu=...) for SMALL calls: where the device path starts to pay.
N_train=250, D=11 (the bench workload), M = 1 ... 65536 rows per call, fp64.

    python tools/small_call_latency.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_emulator_amd import GaussianProcess  # noqa: E402
from bench import synthetic_inputs  # noqa: E402

N, D = 250, 11
inputs, testing, theta, invQ, invQt = synthetic_inputs(1000, N, D, 65536)
gp = GaussianProcess(inputs, [])
gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


sizes = (1, 16, 256, 1024, 4096, 16384, 32768, 65536)
# all the device timings first: the numpy path leaves BLAS worker threads spinning, and on a box
# with a CPU quota they throttle whatever runs next (seen: 4 ms for a 4096-row device call)
gpu = {}
for M in sizes:
    t = testing[:M]
    gpu[M] = timed(lambda: gp.predict(t, is_gpu=True), 200 if M <= 1024 else 40)
print("rows per call   numpy us    gpu us    gpu/numpy")
for M in sizes:
    t = testing[:M]
    c = timed(lambda: gp.predict(t), 20 if M <= 256 else 2)
    print("%9d    %10.1f %9.1f   %8.3f" % (M, c * 1e6, gpu[M] * 1e6, gpu[M] / c))
