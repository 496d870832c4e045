#!/usr/bin/env python3
"""BASELINE config 3 host to host: 2101 per-band emulators on shared inputs (N=250, D=11), 1e5 shared
test rows in a host array, mean / variance / gradient of every band back in host arrays (21.9 GB) through
BatchModel.predict (the slab pipeline).  Reports set-up (pack + upload 2101 emulators) and predict time.

    python tools/c3_host_timing.py [n_emulators] [n_rows]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_emulator_amd import _lib  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 2101
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
N, D = 250, 11
_lib.bind_near_device(0)
rs = np.random.RandomState(3)
inputs, testing = rs.random_sample((N, D)), rs.random_sample((M, D))
thetas, invQts = rs.random_sample((E, D + 2)), rs.random_sample((E, N))
invQs = rs.random_sample((E, N, N))
ctx = _lib.Context(0)
t0 = time.perf_counter()
batch = _lib.BatchModel(ctx, np.exp(thetas), inputs, invQts, invQs, np.float64)
t_setup = time.perf_counter() - t0
out = (np.empty((E, M)), np.empty((E, M)), np.empty((E, M, D)))
for a in out:
    a.fill(0.0)                                    # touch the pages once: the timing below is the steady state
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    batch.predict(testing, out=out)
    ts.append(time.perf_counter() - t0)
gb = sum(a.nbytes for a in out) / 1e9
# spot check of three bands against the numpy path
from gp_emulator_amd import GaussianProcess  # noqa: E402
worst = 0.0
for e in (0, E // 2, E - 1):
    gp = GaussianProcess(inputs, [])
    gp.theta, gp.invQ, gp.invQt = thetas[e], invQs[e], invQts[e]
    ref = gp.predict(testing[:300])
    worst = max(worst, max(float(np.max(np.abs(r - o[e][:300])) / np.max(np.abs(r))) for r, o in zip(ref, out)))
print("C3 host to host: %d emulators x %d rows: set-up (pack + upload) %.2f s; predict %.0f ms (best of 3: %s) = %.2e pairs/s, "
      "%.1f GB of results at %.1f GB/s; parity on 3 bands x 300 rows %.1e" % (
          E, M, t_setup, min(ts) * 1e3, ", ".join("%.0f" % (t * 1e3) for t in ts), E * M / min(ts), gb, gb / min(ts), worst))
