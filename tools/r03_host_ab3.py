import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import GaussianProcess, _lib
from bench import synthetic_inputs
_lib.bind_near_device(0)
def best(fn, reps=7):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3
N, D = int(os.environ.get("GP_AB_N", "250")), 11
inputs, testing, theta, invQ, invQt = synthetic_inputs(1, N, D, 8000000)
gp = GaussianProcess(inputs, []); gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
m = gp.gpu_model(np.float64)
for M in ((1000000, 8000000) if os.environ.get("GP_AB_N") else (250000, 500000, 1000000, 2000000, 4000000, 8000000)):
    out = (np.empty(M), np.empty(M), np.empty((M, D)))
    lo, med = best(lambda: m.predict(testing[:M], out=out), 5)
    print("DIRSTREAMS=%s M=%8d: %.2f ms (median %.2f) = %.3g pts/s" % (os.environ.get("GP_PIPE_DIRSTREAMS", "3"), M, lo, med, M / lo * 1e3))
