#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own numpy path.

Run in the build container only (needs /root/reference; the GPU box has none):

    python tests/golden/make_golden.py

How the reference is executed: its package is Python-2 source, so each module's
text is read from /root/reference, tabs expanded, translated IN MEMORY with
lib2to3 and exec'd into a fresh module object.  Nothing derived from the
reference's source is written to disk; only the numeric inputs/outputs below are
stored.  ``_gpu_predict`` (the CUDA extension the reference imports at module
top, gp_emulator/GaussianProcess.py:7) is replaced by an empty stub module -- the
numpy path never calls it.

Stored vectors (all float64):
  synthetic cases  -- inputs regenerated from ``seed`` by
                      oracle.gp_oracle.benchmark_inputs (the seeded version of
                      tests/benchmark.py:11-15,28-29); outputs mu/var/deriv
                      [/hess] are the reference's cpu_predict / hessian.
  prosail_pc0      -- real emulator data/prosail_30_0_30_0.npz (the fixture of
                      tests/test_perband_emulator.py:10): PC-0 GaussianProcess
                      exactly as MultivariateEmulator builds it
                      (gp_emulator/multivariate_gp.py:175-188), with the
                      reference-computed invQ/invQt stored (LAPACK inverses differ
                      in the last bits across machines and var cancels ~1e7x).
  prosail_mv       -- MultivariateEmulator(dump=...).predict(y) known answers.
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("GP_REFERENCE", "/root/reference")

from oracle import gp_oracle  # noqa: E402  (recipe + self-check only)


def load_reference():
    """Import the reference's gp_emulator modules via in-memory 2to3."""
    warnings.simplefilter("ignore")
    from lib2to3 import refactor
    fixers = [f for f in refactor.get_fixers_from_package("lib2to3.fixes")
              if not f.endswith("fix_import")]   # keep py2 implicit-relative names
    tool = refactor.RefactoringTool(fixers)
    sys.modules["_gpu_predict"] = types.ModuleType("_gpu_predict")  # stub
    mods = {}
    for name in ("GaussianProcess", "multivariate_gp"):
        path = os.path.join(REF, "gp_emulator", name + ".py")
        with open(path) as fh:
            src = fh.read().expandtabs(8)
        if not src.endswith("\n"):
            src += "\n"
        tree = tool.refactor_string(src, path)
        mod = types.ModuleType(name)
        mod.__file__ = path
        sys.modules[name] = mod            # py2 implicit relative imports
        exec(compile(str(tree), path, "exec"), mod.__dict__)
        mods[name] = mod
    return mods


def ref_gp(mods, inputs, theta, invQ, invQt):
    """A reference GaussianProcess set up as tests/benchmark.py:11-15,31-32 does."""
    gp = mods["GaussianProcess"].GaussianProcess(inputs, [])
    gp.D = inputs.shape[1]
    gp.theta = theta
    gp.invQ = invQ
    gp.invQt = invQt
    return gp


SYNTHETIC = [
    # name, seed, N, D, M, M_hess
    ("c1_n100_d5", 101, 100, 5, 10000, 128),
    ("bench_n250_d10", 202, 250, 10, 1500, 0),
    ("c2_n250_d11", 303, 250, 11, 2000, 256),
    ("c4_n300_d11", 404, 300, 11, 1000, 0),
    ("c5_n300_d16", 505, 300, 16, 512, 256),
    ("odd_n37_d3", 606, 37, 3, 333, 33),
    ("one_n1_d1", 707, 1, 1, 5, 5),
]


def main():
    mods = load_reference()
    out = HERE
    for name, seed, N, D, M, MH in SYNTHETIC:
        inputs, testing, theta, invQ, invQt = gp_oracle.benchmark_inputs(seed, N, D, M)
        gp = ref_gp(mods, inputs, theta, invQ, invQt)
        mu, var, deriv = gp.predict(testing, is_gpu=False)
        mu2, deriv2 = gp.predict(testing, do_unc=False)
        assert np.array_equal(mu, mu2) and np.array_equal(deriv, deriv2)
        d = dict(seed=seed, N=N, D=D, M=M, mu=mu, var=var, deriv=deriv)
        if MH:
            d["hess"] = gp.hessian(testing[:MH])
        # self-check of the restatement on the generating machine
        o = gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing)
        print(name, "oracle-vs-reference max abs diff:",
              [float(np.max(np.abs(a - b))) for a, b in zip(o, (mu, var, deriv))],
              "hess" if MH else "",
              float(np.max(np.abs(gp_oracle.hessian(inputs, theta, invQt, testing[:MH])
                                  - d["hess"]))) if MH else "")
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)

    # ---- real emulator (PROSAIL) -------------------------------------------
    npz = os.path.join(REF, "data", "prosail_30_0_30_0.npz")
    mv = mods["multivariate_gp"].MultivariateEmulator(dump=npz)
    f = np.load(npz, allow_pickle=False)
    y_train = f["y"]
    hyper = f["hyperparams"]
    basis = f["basis_functions"]
    train_data = mv.compress(f["X"])                       # (n_pcs, 250)
    rs = np.random.RandomState(909)
    lo, hi = y_train.min(axis=0), y_train.max(axis=0)
    testing = lo + (hi - lo) * rs.random_sample((500, y_train.shape[1]))
    gp0 = mv.emulators[0]
    mu, var, deriv = gp0.predict(testing)
    hess = gp0.hessian(testing[:64])
    np.savez_compressed(
        os.path.join(out, "prosail_pc0.npz"),
        inputs=y_train, targets=train_data[0], theta=hyper[:, 0],
        invQ=gp0.invQ, invQt=gp0.invQt, testing=testing,
        mu=mu, var=var, deriv=deriv, hess=hess)
    pl = gp_oracle.prepare_likelihood(y_train, train_data[0], hyper[:, 0])
    print("prosail_pc0 prepare_likelihood vs reference:",
          float(np.max(np.abs(pl["invQ"] - gp0.invQ))),
          float(np.max(np.abs(pl["invQt"] - gp0.invQt))))

    # ---- training objective (reference :77-125), from the reference's own methods -------
    train = {}
    gpt = mods["GaussianProcess"].GaussianProcess(y_train, train_data[0])
    th = hyper[:, 0]
    train["prosail_pc0_theta"] = th
    train["prosail_pc0_loglik"] = gpt.loglikelihood(th)
    train["prosail_pc0_grad"] = gpt.partial_devs(th)
    rs2 = np.random.RandomState(4242)
    Xs = rs2.random_sample((120, 4))
    ts = np.sin(3 * Xs[:, 0]) + Xs[:, 1] ** 2 - 0.5 * Xs[:, 2] * Xs[:, 3]
    gps = mods["GaussianProcess"].GaussianProcess(Xs, ts)
    thetas = np.array([[0.3, -0.2, 0.1, 0.4, 0.5, -3.0], [1.0, 0.5, -0.5, 0.0, 0.0, -6.0],
                       [-1.0, -1.0, -1.0, -1.0, 1.0, -1.0]])
    train["smooth_inputs"], train["smooth_targets"], train["smooth_thetas"] = Xs, ts, thetas
    # partial_devs(theta) reads the state left by the last loglikelihood/_set_params call and
    # ignores its argument's value for Z/invQ (reference :97-125), as L-BFGS calls them in
    # that order at one point; so evaluate the pair per theta
    ll, gr = [], []
    for t in thetas:
        ll.append(gps.loglikelihood(t))
        gr.append(gps.partial_devs(t))
    train["smooth_loglik"], train["smooth_grad"] = np.array(ll), np.array(gr)
    np.savez_compressed(os.path.join(out, "training_objective.npz"), **train)
    print("training objective vs oracle:",
          float(abs(gp_oracle.loglikelihood(y_train, train_data[0], th) - train["prosail_pc0_loglik"])),
          float(np.max(np.abs(gp_oracle.partial_devs(Xs, ts, thetas[0]) - train["smooth_grad"][0]))))

    # MultivariateEmulator.predict known answers (single test point API,
    # gp_emulator/multivariate_gp.py:195-222)
    pts = np.vstack([y_train[0], y_train[17], testing[0], testing[1]])
    fwd = []
    jac = []
    for p in pts:
        a, b = mv.predict(p)
        fwd.append(np.asarray(a))
        jac.append(np.asarray(b))
    np.savez_compressed(
        os.path.join(out, "prosail_mv.npz"),
        y_train=y_train, hyperparams=hyper, basis_functions=basis,
        train_data=train_data, n_pcs=int(f["n_pcs"]), points=pts,
        fwd=np.array(fwd), jac=np.array(jac), x_train_row0=f["X"][0],
        x_train_row17=f["X"][17])
    print("mv.predict(y_train[0]) vs X_train[0]: max abs",
          float(np.max(np.abs(fwd[0] - f["X"][0]))))


if __name__ == "__main__":
    main()
