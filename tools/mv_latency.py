#!/usr/bin/env python3
"""Latency of the MultivariateEmulator calls an optimiser makes one state vector at a time
(reference multivariate_gp.py:195-222; EO-LDAS calls predict once per cost evaluation):
predict(y) on the numpy path, predict(y, is_gpu=True), and predict_many for 1 / 16 / 1024 rows.

    python tools/mv_latency.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_emulator_amd import MultivariateEmulator  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "prosail_mv.npz"))
mv = MultivariateEmulator(X=g["train_data"].T @ g["basis_functions"], y=g["y_train"], hyperparams=g["hyperparams"],
                          basis_functions=g["basis_functions"], n_pcs=int(g["n_pcs"]), is_gpu=True)
rs = np.random.RandomState(3)
lo, hi = mv.y_train.min(0), mv.y_train.max(0)
Y = lo + (hi - lo) * rs.random_sample((1024, lo.size))


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


t_cpu, ref = timed(lambda: mv.predict(Y[0]), 20)
t_gpu, got = timed(lambda: mv.predict(Y[0], is_gpu=True), 200)
print("%d PCs, N=%d, D=%d, %d bands" % (mv.n_pcs, mv.y_train.shape[0], lo.size, mv.basis_functions.shape[1]))
print("predict(y)               numpy  %8.1f us" % (t_cpu * 1e6))
print("predict(y, is_gpu=True)         %8.1f us   max|fwd - numpy| %.1e  max|deriv - numpy| %.1e" % (
    t_gpu * 1e6, np.max(np.abs(got[0] - ref[0])), np.max(np.abs(got[1] - ref[1]))))
for m in (1, 16, 1024):
    t, out = timed(lambda: mv.predict_many(Y[:m], do_deriv=True), 50 if m < 1024 else 10)
    print("predict_many(%4d rows, do_deriv) %8.1f us  = %6.1f us per row" % (m, t * 1e6, t * 1e6 / m))
# the per-call content check (digest of every emulator's constants and the basis, taken while the device works)
from gp_emulator_amd import _lib  # noqa: E402
blocks = list(mv._gpu.values())[0]["blocks"]
t0 = time.perf_counter()
for _ in range(200):
    blocks.digest()
print("digest of the %.1f MB the resident copy was made from, alone on the calling thread: %.1f us"
      % (sum(blocks.lens) / 1e6, (time.perf_counter() - t0) / 200 * 1e6))
