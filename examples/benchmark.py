#!/usr/bin/env python3
"""Python-3 counterpart of the reference's tests/benchmark.py (same flow, same sizes, same
pass mark), running against gp_emulator_amd:

  for Npredict in 1e5 .. 9e5 step 1e5 (Ntrain = 250, Ninputs = 10, U[0,1) inputs, theta,
  invQ, invQt -- tests/benchmark.py:11-15,24-32):
      CPU:  gp.predict(testing, is_gpu=False)                                     (:36)
      GPU:  gp.predict(testing, is_gpu=True, precision=np.float32, threshold=1e5) (:42)
      pass iff max-norm relative errors of mu, var, deriv are all <= 1e-5         (:51-59)

Both legs are timed host arrays in -> host arrays out, as the reference does (:35-44).
The reference's unseeded np.random is replaced by a seeded RandomState so runs repeat.

    python examples/benchmark.py [--max 9e5] [--precision float32|float64]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import GaussianProcess  # noqa: E402


def set_testing_val(gp, rs, Ninputs, Ntrain):      # tests/benchmark.py:11-15
    gp.D = Ninputs
    gp.theta = rs.random_sample(Ninputs + 2)
    gp.invQ = rs.random_sample((Ntrain, Ntrain))
    gp.invQt = rs.random_sample(Ntrain)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max", type=float, default=9e5)
    ap.add_argument("--precision", default="float32", choices=["float32", "float64"])
    a = ap.parse_args()
    precision = np.dtype(a.precision).type
    tol = 1e-5
    print("Problem_size\tCPU time\tGPU time\tSpeedup\tStatus")
    print("-----------------------------")
    rs = np.random.RandomState(2013)
    failed = 0
    for Npredict in range(int(1e5), int(a.max) + 1, int(1e5)):
        Ntrain, Ninputs = 250, 10
        inputs = rs.random_sample((Ntrain, Ninputs))
        testing = rs.random_sample((Npredict, Ninputs))
        gp = GaussianProcess(inputs, [])
        set_testing_val(gp, rs, Ninputs, Ntrain)

        start = time.time()
        mu_c, var_c, deriv_c = gp.predict(testing, is_gpu=False)
        cputime = time.time() - start

        start = time.time()
        mu_g, var_g, deriv_g = gp.predict(testing, is_gpu=True, precision=precision, threshold=1e5)
        gputime = time.time() - start

        e_mu = max(abs(mu_c - mu_g)) / np.max(abs(mu_c))
        e_var = max(abs(var_c - var_g)) / np.max(abs(var_c))
        e_deriv = np.max(abs(deriv_c - deriv_g)) / np.max(abs(deriv_c))
        ok = not (e_mu > tol or e_var > tol or e_deriv > tol)
        failed += not ok
        print("%d\t%.2fs\t%.4fs\t%.0fx\t%s\te_mu=%.2g\te_var=%.2g\te_deriv=%.2g"
              % (Npredict, cputime, gputime, cputime / gputime, "Pass" if ok else "FAILED",
                 e_mu, e_var, e_deriv), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
