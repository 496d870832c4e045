"""Soak: many small host-path calls of varying shape/precision; checks results stay right and
per-call time does not drift (a leak of device memory or pinned buffers would show)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import GaussianProcess
from bench import synthetic_inputs

rs = np.random.RandomState(0)
shapes = [(250, 11), (100, 5), (300, 16), (37, 3), (400, 4)]
cases = []
for N, D in shapes:
    inputs, testing, theta, invQ, invQt = synthetic_inputs(N + D, N, D, 70000)
    gp = GaussianProcess(inputs, [])
    gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
    ref = gp.predict(testing[:512], is_gpu=False)          # the API's explicit numpy branch
    cases.append((gp, testing, ref))
t_first = t_last = None
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for it in range(n_iter):
    gp, testing, ref = cases[it % len(cases)]
    prec = np.float64 if (it // len(cases)) % 2 == 0 else np.float32
    M = [512, 5000, 70000][it % 3]
    t0 = time.perf_counter()
    out = gp.predict(testing[:M], is_gpu=True, precision=prec, threshold=2e5)
    if it % 7 == 0:
        h = gp.hessian(testing[:64], is_gpu=True, precision=prec) if gp.D <= 16 else None
    dt = time.perf_counter() - t0
    tol = 1e-10 if prec == np.float64 else 1e-4
    e = max(float(np.max(np.abs(r - o[:512])) / np.max(np.abs(r))) for r, o in zip(ref, out))
    assert e <= tol, (it, e)
    if it == 50:
        t_first = time.perf_counter()
    if it == n_iter - 1:
        t_last = time.perf_counter()
print("soak OK: %d calls; mean %.2f ms per call after warm-up" % (n_iter, (t_last - t_first) / (n_iter - 51) * 1e3))
