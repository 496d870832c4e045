"""A/B inside one process is not possible (the switch is read once): run this twice with GP_PIPE_DIRSTREAMS=0/1."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import GaussianProcess, _lib, multi_gpu
from bench import synthetic_inputs
_lib.bind_near_device(0)
def best(fn, reps=7):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3
N, D, M = 250, 11, 1000000
inputs, testing, theta, invQ, invQt = synthetic_inputs(1, N, D, M)
gp = GaussianProcess(inputs, []); gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
for prec in (np.float64, np.float32):
    print("DIRSTREAMS=%s f64 rows, precision %s: %.2f ms (median %.2f)" % ((os.environ.get("GP_PIPE_DIRSTREAMS", "1"), np.dtype(prec).name) + best(lambda: gp.predict(testing, is_gpu=True, precision=prec))))
N, D, M = 300, 11, 12500000
inputs, testing, theta, invQ, invQt = synthetic_inputs(7, N, D, M)
gp = GaussianProcess(inputs, []); gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
out = (np.empty(M), np.empty(M), np.empty((M, D)))
print("DIRSTREAMS=%s C4 shard reused outputs: %.1f ms (median %.1f)" % ((os.environ.get("GP_PIPE_DIRSTREAMS", "1"),) + best(lambda: multi_gpu.predict_sharded(gp, testing, [0], out=out), 4)))
