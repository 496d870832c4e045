"""Pin the oracle: the numpy restatement must reproduce the vectors the REFERENCE
produced (tests/golden/make_golden.py).  CPU only.

Tolerance: bit-identical on the generating machine; 1e-13 max-norm relative
elsewhere (OpenBLAS kernels differ per CPU, so dot-product summation order may).
Real-data variance is compared relative to b = exp(theta[D]) because
var = b - k^T invQ k cancels ~1e7-fold there.
"""
import numpy as np
import pytest

from conftest import SYNTHETIC_CASES, load_golden, synthetic_case
from oracle import gp_oracle

TOL = 1e-13


@pytest.mark.parametrize("name", SYNTHETIC_CASES)
def test_cpu_predict_matches_reference(name):
    g = synthetic_case(name)
    mu, var, deriv = gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"],
                                           g["invQt"], g["testing"])
    assert mu.shape == g["mu"].shape and deriv.shape == g["deriv"].shape
    assert gp_oracle.maxnorm_err(g["mu"], mu) <= TOL
    assert gp_oracle.maxnorm_err(g["var"], var) <= TOL
    assert gp_oracle.maxnorm_err(g["deriv"], deriv) <= TOL
    mu2, deriv2 = gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"],
                                        g["invQt"], g["testing"], do_unc=False)
    assert np.array_equal(mu, mu2) and np.array_equal(deriv, deriv2)


@pytest.mark.parametrize("name", [n for n in SYNTHETIC_CASES])
def test_hessian_matches_reference(name):
    g = synthetic_case(name)
    if "hess" not in g:
        pytest.skip("no hessian stored for this case")
    mh = g["hess"].shape[0]
    h = gp_oracle.hessian(g["inputs"], g["theta"], g["invQt"], g["testing"][:mh])
    assert h.shape == g["hess"].shape
    assert gp_oracle.maxnorm_err(g["hess"], h) <= TOL


def test_prosail_pc0_real_emulator():
    g = load_golden("prosail_pc0")
    b = float(np.exp(g["theta"][g["inputs"].shape[1]]))
    mu, var, deriv = gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"],
                                           g["invQt"], g["testing"])
    assert gp_oracle.maxnorm_err(g["mu"], mu) <= 1e-11
    assert gp_oracle.maxnorm_err(g["deriv"], deriv) <= 1e-11
    assert np.max(np.abs(var - g["var"])) / b <= 1e-8
    h = gp_oracle.hessian(g["inputs"], g["theta"], g["invQt"], g["testing"][:64])
    assert gp_oracle.maxnorm_err(g["hess"], h) <= 1e-11
    # the producer of the hot path's inputs (GaussianProcess.py:52-75); LAPACK
    # inverse of a cond~1e7 matrix: compare loosely
    pl = gp_oracle.prepare_likelihood(g["inputs"], g["targets"], g["theta"])
    assert np.max(np.abs(pl["invQ"] - g["invQ"])) / np.max(np.abs(g["invQ"])) <= 1e-6
    assert np.max(np.abs(pl["invQt"] - g["invQt"])) / np.max(np.abs(g["invQt"])) <= 1e-6


def test_gradient_and_hessian_are_derivatives():
    """Known-answer property the reference has no test for: deriv/hessian agree
    with central finite differences of mu."""
    g = synthetic_case("odd_n37_d3")
    x = g["testing"][:8]
    mu, var, deriv = gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"], g["invQt"], x)
    hess = gp_oracle.hessian(g["inputs"], g["theta"], g["invQt"], x)
    h = 1e-5
    for d in range(x.shape[1]):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += h
        xm[:, d] -= h
        mp, _, dp = gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"], g["invQt"], xp)
        mm, _, dm = gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"], g["invQt"], xm)
        assert np.allclose((mp - mm) / (2 * h), deriv[:, d], rtol=1e-6, atol=1e-7)
        assert np.allclose((dp - dm) / (2 * h), hess[:, d, :], rtol=1e-6, atol=1e-6)


# (size, block_size) -> (starts, ends); hand-derived from
# gp_emulator/GaussianProcess.py:253-270 under Python-2 integer division (:265)
BLOCK_CASES = [
    ((250000, 100000), ([0, 100000, 175000], [100000, 175000, 250000])),
    ((200000, 100000), ([0, 100000], [100000, 200000])),
    ((1000, 200000), ([0], [1000])),
    ((200001, 100000), ([0, 100000, 150000], [100000, 150000, 200001])),
    ((300001, 100000), ([0, 100000, 200000, 250000], [100000, 200000, 250000, 300001])),
    ((7, 3), ([0, 3, 5], [3, 5, 7])),
]


@pytest.mark.parametrize("args,expect", BLOCK_CASES)
def test_get_gpu_block(args, expect):
    s, e = gp_oracle.get_gpu_block(*args)
    assert list(s) == expect[0] and list(e) == expect[1]
    assert s[0] == 0 and e[-1] == args[0] and np.all(s[1:] == e[:-1])


def test_multivariate_known_answer():
    """MultivariateEmulator.predict(y_train[i]) reproduces X_train[i] to ~1e-3
     and matches the reference's own outputs."""
    g = load_golden("prosail_mv")
    emus = []
    for i in range(int(g["n_pcs"])):
        pl = gp_oracle.prepare_likelihood(g["y_train"], g["train_data"][i], g["hyperparams"][:, i])
        emus.append((g["y_train"], g["hyperparams"][:, i], pl["invQ"], pl["invQt"]))
    for j, p in enumerate(g["points"]):
        fwd, jac = gp_oracle.multivariate_predict(g["basis_functions"], emus, p)
        assert np.max(np.abs(fwd - g["fwd"][j])) <= 1e-7
        assert np.max(np.abs(jac - g["jac"][j])) / np.max(np.abs(g["jac"][j])) <= 1e-6
    fwd0, _ = gp_oracle.multivariate_predict(g["basis_functions"], emus, g["points"][0])
    assert np.max(np.abs(fwd0 - g["x_train_row0"])) <= 2e-3


def test_training_objective_matches_reference():
    """loglikelihood / partial_devs restatement against the reference's own methods
    (gp_emulator/GaussianProcess.py:77-125; goldens in training_objective.npz)."""
    g = load_golden("training_objective")
    for k in range(len(g["smooth_thetas"])):
        ll = gp_oracle.loglikelihood(g["smooth_inputs"], g["smooth_targets"], g["smooth_thetas"][k])
        gr = gp_oracle.partial_devs(g["smooth_inputs"], g["smooth_targets"], g["smooth_thetas"][k])
        assert abs(ll - g["smooth_loglik"][k]) <= 1e-9 * abs(g["smooth_loglik"][k])
        assert np.max(np.abs(gr - g["smooth_grad"][k])) <= 1e-7 * np.max(np.abs(g["smooth_grad"][k]))
    p = load_golden("prosail_pc0")
    ll = gp_oracle.loglikelihood(p["inputs"], p["targets"], g["prosail_pc0_theta"])
    assert abs(ll - g["prosail_pc0_loglik"]) <= 1e-8 * abs(g["prosail_pc0_loglik"])
    # the gradient is the derivative of the cost (central differences)
    th = g["smooth_thetas"][0]
    gr = gp_oracle.partial_devs(g["smooth_inputs"], g["smooth_targets"], th)
    for d in range(len(th)):
        tp, tm = th.copy(), th.copy()
        tp[d] += 1e-5
        tm[d] -= 1e-5
        fd = (gp_oracle.loglikelihood(g["smooth_inputs"], g["smooth_targets"], tp) -
              gp_oracle.loglikelihood(g["smooth_inputs"], g["smooth_targets"], tm)) / 2e-5
        assert abs(fd - gr[d]) <= 1e-5 * max(1.0, abs(gr[d]))
