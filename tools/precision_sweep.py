"""fp32 vs fp64 tolerance sweep (BASELINE configs[4]: "fp32 vs fp64 tolerance sweep"): max-norm
relative error (tests/benchmark.py:51-53) of mean, variance, gradient and Hessian of the GPU path
in both precisions against the API's explicit numpy branch (float64), over

  * training-set size N in {100, 250, 300} and input dimension D in {5, 11, 16},
  * the lengthscale: theta[:D] shifted by log(scale), scale in {1/16, 1/4, 1, 4, 16}
    (e = exp(theta) multiplies squared distances: small scale = long lengthscale = smooth, badly
    conditioned Q; large scale = short lengthscale = nearly diagonal Q),
  * two kinds of emulator: "bench" = the reference benchmark's random invQ / invQt
    (tests/benchmark.py:11-15, not a real inverse) and "gp" = a trained-like GP (smooth targets,
    invQ / invQt from _prepare_likelihood, noise e[D+1] = 1e-6) whose variance cancels like a real one,
  * the real PROSAIL emulator (tests/golden/prosail_pc0.npz: N=250, D=10, cond(Q) ~ 3.5e7).

float32 runs take the caller's float64 rows (constants packed from float64, rows centred and
scaled in double while staged).  The variance of a GP is judged as |dvar| / b as well.

    python tools/precision_sweep.py > profiles/r02_precision_sweep.txt
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import GaussianProcess


def err(ref, got):
    return float(np.max(np.abs(np.asarray(ref) - np.asarray(got))) / np.max(np.abs(ref)))


def run(gp, testing, label):
    ref = gp.predict(testing, is_gpu=False)
    href = gp.hessian(testing[:256], is_gpu=False)
    b = float(np.exp(gp.theta[gp.D]))
    row = [label]
    for prec in (np.float64, np.float32):
        got = gp.predict(testing, is_gpu=True, precision=prec)
        h = gp.hessian(testing[:256], is_gpu=True, precision=prec)
        row += ["%.1e" % err(ref[0], got[0]), "%.1e" % err(ref[1], got[1]),
                "%.1e" % (np.max(np.abs(ref[1] - got[1])) / b), "%.1e" % err(ref[2], got[2]),
                "%.1e" % err(href, h)]
    print(" | ".join(row), flush=True)


def main():
    print("# columns: case | fp64: e_mu e_var |dvar|/b e_grad e_hess | fp32: e_mu e_var |dvar|/b e_grad e_hess")
    print("# (max-norm relative errors against the numpy float64 branch; 2000 test rows, Hessian on 256)")
    M = 2000
    for N, D in ((100, 5), (250, 11), (300, 11), (300, 16)):
        for scale in (1 / 16., 1 / 4., 1., 4., 16.):
            rs = np.random.RandomState(100 + N + D)
            inputs, testing = rs.random_sample((N, D)), rs.random_sample((M, D))
            theta = rs.random_sample(D + 2)
            theta[:D] += np.log(scale)
            # bench: random non-symmetric invQ, as the reference benchmark feeds
            gp = GaussianProcess(inputs, [])
            gp.theta, gp.invQ, gp.invQt = theta, rs.random_sample((N, N)), rs.random_sample(N)
            run(gp, testing, "bench N=%d D=%d scale=%-6g" % (N, D, scale))
            # gp: smooth targets, real inverse
            targets = np.sin(inputs @ np.linspace(0.5, 2.0, D)) + 0.1 * inputs[:, 0] ** 2
            gp = GaussianProcess(inputs, targets)
            th = theta.copy()
            th[D + 1] = np.log(1e-6)
            gp._set_params(th)
            run(gp, testing, "gp    N=%d D=%d scale=%-6g cond(Q)=%.1e" % (N, D, scale, np.linalg.cond(gp.Q)))
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                             "tests", "golden", "prosail_pc0.npz"), allow_pickle=False)
    gp = GaussianProcess(g["inputs"], [])
    gp.theta, gp.invQ, gp.invQt = g["theta"], g["invQ"], g["invQt"]
    run(gp, g["testing"], "PROSAIL pc0 N=250 D=10 (real emulator, stored reference invQ)")
    # the float32 floor of that emulator, in numpy: everything float64 except that the kernel row
    # k_i (resp. the exponent argument) is rounded to float32 once
    import scipy.spatial.distance as dist
    e = np.exp(g["theta"])
    D = g["inputs"].shape[1]
    sq = np.sqrt(e[:D])
    arg = -0.5 * dist.cdist(sq * g["inputs"], sq * g["testing"], "sqeuclidean")
    k = e[D] * np.exp(arg)
    mu = k.T @ g["invQt"]
    cond = np.median((np.abs(k) * np.abs(g["invQt"])[:, None]).sum(axis=0) / np.abs(mu))
    k32 = k.astype(np.float32).astype(np.float64)
    a32 = e[D] * np.exp(arg.astype(np.float32).astype(np.float64))
    print("PROSAIL float32 floor (numpy): condition number of the mean's sum, median over rows: %.2g; "
          "e_mu with k rounded to float32: %.1e; with the exponent argument rounded to float32: %.1e"
          % (cond, err(mu, k32.T @ g["invQt"]), err(mu, a32.T @ g["invQt"])))


if __name__ == "__main__":
    main()
