#!/usr/bin/env python3
"""Host to host predict with the caller's arrays page-locked (gp_emulator_amd.pinned_empty) against the same call on
pageable arrays (the staged slab pipeline): 1e6 rows, N=250, D=11, fp64 and fp32.

    python tools/pinned_path_timing.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gp_emulator_amd
from gp_emulator_amd import GaussianProcess, _lib
from bench import synthetic_inputs

print("bound to %d cpus next to device 0" % len(_lib.bind_near_device(0)))
N, D, M = 250, 11, 1000000
inputs, testing, theta, invQ, invQt = synthetic_inputs(1, N, D, M)
gp = GaussianProcess(inputs, [])
gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt


def best(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), float(np.median(ts))


for prec in (np.float64, np.float32):
    m = gp.gpu_model(prec)
    rows = testing.astype(prec)
    out = m.predict(rows)
    out = tuple(np.array(a) for a in out)
    lo, med = best(lambda: m.predict(rows, out=out))
    print("%s pageable arrays (staged pipeline): %.2f ms (median %.2f) -> %.3g pts/s" % (np.dtype(prec).name, lo * 1e3, med * 1e3, M / lo))
    t_pin = gp_emulator_amd.pinned_empty((M, D), prec)
    t_pin[...] = rows
    pout = (gp_emulator_amd.pinned_empty((M,), prec), gp_emulator_amd.pinned_empty((M,), prec), gp_emulator_amd.pinned_empty((M, D), prec))
    lo, med = best(lambda: m.predict(t_pin, out=pout))
    same = all(np.array_equal(a, b) for a, b in zip(out, pout))
    isz = np.dtype(prec).itemsize
    print("%s page-locked arrays (no staging):   %.2f ms (median %.2f) -> %.3g pts/s, %.1f GB/s over the link, same bits: %s"
          % (np.dtype(prec).name, lo * 1e3, med * 1e3, M / lo, M * (2 * D + 2) * isz / lo / 1e9, same))
