#!/usr/bin/env python3
"""Python-3 counterpart of the reference's tests/test_perband_emulator.py flow:

  emulators = MultivariateEmulator(dump=prosail npz)                    (:11)
  per-band emulator: a GaussianProcess on y_train whose targets are the band mean of
      X_train over columns 442:476, hyper-parameters learnt with n_tries=5     (:22-37,42-44)
  gpu vs cpu predict on validation rows, difference printed                    (:47-51)

The reference's data files are not shipped here: X_train is rebuilt from the committed fixture
(tests/golden/prosail_mv.npz: PC weights x basis functions, i.e. the 99 %-variance
reconstruction) unless --npz points at data/prosail_30_0_30_0.npz, and the missing
validation file (.MISSING_LARGE_BLOBS:1) is replaced by seeded uniform rows inside the
per-column range of y_train.

    python examples/perband_emulator.py [--npz path] [--tries 5] [--cpu-learn]
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_emulator_amd import GaussianProcess  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--npz", default=None)
    ap.add_argument("--tries", type=int, default=5)
    ap.add_argument("--cpu-learn", action="store_true", help="also time the numpy learning branch")
    a = ap.parse_args()
    if a.npz:
        with np.load(a.npz, allow_pickle=False) as f:
            X_train, y_train = f["X"], f["y"]
    else:
        with np.load(os.path.join(ROOT, "tests", "golden", "prosail_mv.npz"), allow_pickle=False) as f:
            X_train, y_train = f["train_data"].T @ f["basis_functions"], f["y_train"]
    band_pass = np.zeros((1, X_train.shape[1]), dtype=bool)
    band_pass[:, 442:476] = True                                       # :42-43
    targets = X_train[:, band_pass[0]].mean(axis=1)                    # :26-27
    rs = np.random.RandomState(1000)
    lo, hi = y_train.min(axis=0), y_train.max(axis=0)
    X = lo + (hi - lo) * rs.random_sample((1000, y_train.shape[1]))    # stands in for d[:, :10]

    gp = GaussianProcess(y_train * 1, targets)
    np.random.seed(5)
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cost, theta = gp.learn_hyperparameters(n_tries=a.tries, is_gpu=True)     # :34
    t_gpu = time.time() - t0
    print("learn_hyperparameters(n_tries=%d, is_gpu=True): cost %.6f in %.2f s" % (a.tries, cost, t_gpu))
    if a.cpu_learn:
        gp_c = GaussianProcess(y_train * 1, targets)
        np.random.seed(5)
        t0 = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cost_c, _ = gp_c.learn_hyperparameters(n_tries=a.tries)
        print("learn_hyperparameters(n_tries=%d, numpy):       cost %.6f in %.2f s" % (a.tries, cost_c, time.time() - t0))
    gpur = gp.predict(X, is_gpu=True)[0]                                # :49
    cpur = gp.predict(X, is_gpu=False)[0]                               # :50
    print("max |gpu - cpu| of the predicted band mean over %d rows: %.3g (values in [%.3g, %.3g])"
          % (len(X), np.max(np.abs(gpur - cpur)), cpur.min(), cpur.max()))
    print("predict finish")
    return 0


if __name__ == "__main__":
    sys.exit(main())
