import sys, time, warnings
sys.path.insert(0, '/root/repo' if __import__('os').path.exists('/root/repo/bench.py') else '.')
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from conftest import load_golden
from gp_emulator_amd import GaussianProcess
g = load_golden("training_objective")
noisy = g["smooth_targets"] + 0.05 * np.random.RandomState(1).standard_normal(120)
for is_gpu in (False, True):
    gp = GaussianProcess(g["smooth_inputs"], noisy)
    np.random.seed(1)
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cost, theta = gp.learn_hyperparameters(n_tries=8, is_gpu=is_gpu)
    print("is_gpu=%s: cost %.9f in %.2f s" % (is_gpu, cost, time.time() - t0), flush=True)
