"""End-to-end (host numpy arrays in -> host numpy arrays out) timing of the drop-in path,
i.e. what tests/benchmark.py:41-44 times, next to the kernel-only rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import GaussianProcess, _lib
from bench import synthetic_inputs

N, D = 250, 11
for M in (100000, 1000000):
    inputs, testing, theta, invQ, invQt = synthetic_inputs(1, N, D, M)
    gp = GaussianProcess(inputs, [])
    gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
    for prec in (np.float64, np.float32):
        for thr in (2e5, 1e7):
            gp.predict(testing[:1000], is_gpu=True, precision=prec, threshold=thr)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                out = gp.predict(testing, is_gpu=True, precision=prec, threshold=thr)
                ts.append(time.perf_counter() - t0)
            print("M=%d %s threshold=%g: predict(is_gpu=True) %.1f ms -> %.3g pts/s" % (
                M, np.dtype(prec).name, thr, min(ts) * 1e3, M / min(ts)), flush=True)
        m = gp.gpu_model(prec)
        m.predict(testing[:1000])
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            out = m.predict(testing)
            ts.append(time.perf_counter() - t0)
        print("M=%d %s Model.predict (row-major, one slab): %.1f ms -> %.3g pts/s" % (
            M, np.dtype(prec).name, min(ts) * 1e3, M / min(ts)), flush=True)
