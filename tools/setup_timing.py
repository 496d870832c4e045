#!/usr/bin/env python3
"""Disk -> ready-to-predict time of a stored multivariate emulator:
MultivariateEmulator(...) with the stored hyper-parameters, host LAPACK inverses (the reference's
route, multivariate_gp.py:185-186) against one launch of the HIP likelihood kernel for all PCs.
Uses the PROSAIL-shaped fixture tests/golden/prosail_mv.npz (N_train=250, D=10, 12 PCs)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_emulator_amd import MultivariateEmulator  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "prosail_mv.npz"))
X = g["train_data"].T @ g["basis_functions"]
kw = dict(X=X, y=g["y_train"], hyperparams=g["hyperparams"], basis_functions=g["basis_functions"],
          n_pcs=int(g["n_pcs"]))
MultivariateEmulator(is_gpu=True, **kw)          # library load, context
for flag in (False, True):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        mv = MultivariateEmulator(is_gpu=flag, **kw)
        best = min(best, time.perf_counter() - t0)
    print("set up 12 PC emulators (N=250, D=10), is_gpu=%s: %.1f ms" % (flag, best * 1e3))

# the same emulator trained from scratch (hyperparams=None: learn_hyperparameters for every PC,
# reference multivariate_gp.py:180-184), all PCs and restarts together on the GPU
import warnings
warnings.simplefilter("ignore")
kw2 = dict(X=X, y=g["y_train"], basis_functions=g["basis_functions"], n_pcs=int(g["n_pcs"]), n_tries=5)
np.random.seed(0)
t0 = time.perf_counter()
mv = MultivariateEmulator(is_gpu=True, **kw2)
print("train 12 PC emulators x 5 restarts (N=250, D=10), is_gpu=True: %.2f s" % (time.perf_counter() - t0))
