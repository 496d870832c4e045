"""CPU oracle for the GP predict hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a Python-3 / numpy restatement of the reference's numpy path
(UCL/gp_emulator, ``gp_emulator/GaussianProcess.py``).  It is the checker the
HIP path is compared with.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; nothing under
``gp_emulator_amd/`` does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the reference's
own ``GaussianProcess`` class in the build container (Python-2 source translated
in memory with ``lib2to3``; nothing of it is stored in this repo) and recorded
its outputs in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against those vectors (bit-for-bit on the machine that made them,
<= 1e-13 relative elsewhere because BLAS summation order may differ).

Every function cites the reference lines it follows (paths relative to the
reference checkout).  Third-party arithmetic used exactly as the reference uses
it: ``scipy.spatial.distance.cdist(..., 'sqeuclidean')``, ``numpy.dot``,
``numpy.exp``, ``numpy.linalg.inv`` (reference pins no versions, README.rst:11-15).
"""
import numpy as np
import scipy.spatial.distance as dist


def prepare_likelihood(inputs, targets, theta):
    """Producer of the hot path's inputs: Q, invQ, invQt, logdetQ.

    Follows gp_emulator/GaussianProcess.py:52-75 (``_prepare_likelihood``) and
    :127-139 (``_set_params``): Z built one input dimension at a time from
    ``np.tile`` differences, ``Q = Z + e[D+1] I``, ``invQ = inv(Q)``,
    ``invQt = invQ . targets``.
    """
    inputs = np.asarray(inputs)
    n, D = inputs.shape
    exp_theta = np.exp(theta)
    Z = np.zeros((n, n))
    for d in range(D):
        Z = Z + exp_theta[d] * ((np.tile(inputs[:, d], (n, 1)) -
                                 np.tile(inputs[:, d], (n, 1)).T)) ** 2
    Z = exp_theta[D] * np.exp(-0.5 * Z)
    Q = Z + exp_theta[D + 1] * np.eye(n)
    invQ = np.linalg.inv(Q)
    invQt = np.dot(invQ, targets)
    logdetQ = 2.0 * np.sum(np.log(np.diag(np.linalg.cholesky(Q))))
    return dict(Z=Z, Q=Q, invQ=invQ, invQt=invQt, logdetQ=logdetQ)


def loglikelihood(inputs, targets, theta):
    """Negative log marginal likelihood as the reference defines its cost,
    gp_emulator/GaussianProcess.py:77-95: 0.5 logdetQ + 0.5 t.invQt + 0.5 n log(2 pi)."""
    pl = prepare_likelihood(inputs, targets, theta)
    n = np.asarray(inputs).shape[0]
    return (0.5 * pl["logdetQ"] + 0.5 * np.dot(targets, pl["invQt"]) +
            0.5 * n * np.log(2. * np.pi))


def partial_devs(inputs, targets, theta):
    """Gradient of that cost w.r.t. the D+2 hyper-parameters,
    gp_emulator/GaussianProcess.py:97-125 (V built with np.tile exactly as there)."""
    inputs = np.asarray(inputs)
    n, D = inputs.shape
    pl = prepare_likelihood(inputs, targets, theta)
    Z, invQ, invQt = pl["Z"], pl["invQ"], pl["invQt"]
    partial_d = np.zeros(D + 2)
    for d in range(D):
        V = (((np.tile(inputs[:, d], (n, 1)) -
               np.tile(inputs[:, d], (n, 1)).T)) ** 2).T * Z
        partial_d[d] = np.exp(theta[d]) * \
            (np.dot(invQt, np.dot(V, invQt)) - np.sum(invQ * V)) / 4.
    partial_d[D] = 0.5 * np.sum(invQ * Z) - 0.5 * np.dot(invQt, np.dot(Z, invQt))
    partial_d[D + 1] = 0.5 * np.trace(invQ) * np.exp(theta[D + 1]) - \
        0.5 * np.dot(invQt, invQt) * np.exp(theta[D + 1])
    return partial_d


def kernel_rows(inputs, theta, testing):
    """K_*^T, shape (N_train, N_test).  gp_emulator/GaussianProcess.py:230-234."""
    D = inputs.shape[1]
    expX = np.exp(theta)
    a = dist.cdist(np.sqrt(expX[:D]) * inputs, np.sqrt(expX[:D]) * testing,
                   'sqeuclidean')
    return expX[D] * np.exp(-0.5 * a)


def cpu_predict(inputs, theta, invQ, invQt, testing, do_unc=True):
    """mean, variance, gradient.  gp_emulator/GaussianProcess.py:211-251.

    Returns ``(mu, var, deriv)`` or ``(mu, deriv)`` when ``do_unc`` is false,
    exactly as the reference does (:248-251).
    """
    nn, D = testing.shape
    assert D == inputs.shape[1]
    expX = np.exp(theta)
    a = kernel_rows(inputs, theta, testing)          # :232-234
    b = expX[D]
    mu = np.dot(a.T, invQt)                          # :237
    if do_unc:
        var = b - np.sum(a * np.dot(invQ, a), axis=0)  # :240
    deriv = np.zeros((nn, D))
    for d in range(D):                               # :244-247
        aa = inputs[:, d].flatten()[None, :] - testing[:, d].flatten()[:, None]
        c = a * aa.T
        deriv[:, d] = expX[d] * np.dot(c.T, invQt)
    if do_unc:
        return mu, var, deriv
    return mu, deriv


def hessian(inputs, theta, invQt, testing):
    """(nn, D, D) Hessian of the mean.  gp_emulator/GaussianProcess.py:345-366."""
    nn, D = testing.shape
    assert D == inputs.shape[1]
    expX = np.exp(theta)
    a = kernel_rows(inputs, theta, testing)          # :352-354
    dd_addition = np.identity(D) * expX[:D]          # :355
    hess = np.zeros((nn, D, D))
    for d in range(D):                               # :357-365
        for d2 in range(D):
            aa = expX[d] * (inputs[:, d].flatten()[None, :] -
                            testing[:, d].flatten()[:, None]) * \
                expX[d2] * (inputs[:, d2].flatten()[None, :] -
                            testing[:, d2].flatten()[:, None]) - \
                dd_addition[d, d2]
            cc = a * (aa.T)
            hess[:, d, d2] = np.dot(cc.T, invQt)
    return hess


def get_gpu_block(size, block_size):
    """Row-block boundaries.  gp_emulator/GaussianProcess.py:253-270.

    The reference is Python 2: ``/`` at :265 is integer division and ``range``
    returns a list that is mutated at :267.  Restated with ``//`` and arrays.
    """
    ind_start = np.array(list(range(int(0), int(size), int(block_size))),
                         dtype=np.int64)
    ind_end = np.append(ind_start[1:], int(size)).astype(np.int64)
    nblocks = len(ind_start)
    if nblocks > 1:
        last_two_block_size = (ind_end[nblocks - 1] - ind_start[nblocks - 2]) // 2
        ind_end[nblocks - 2] = ind_start[nblocks - 2] + last_two_block_size
        ind_start[nblocks - 1] = ind_end[nblocks - 2]
    assert np.all(ind_end - ind_start <= block_size)
    return ind_start, ind_end


def cpu_predict_blocked(inputs, theta, invQ, invQt, testing, block=100000):
    """``cpu_predict`` over row blocks (bounded temporaries, 5 x N x block x 8 B).

    Rows are independent (gp_emulator/GaussianProcess.py:232-247), so blocking
    changes no arithmetic except BLAS-internal summation order.  Used by the
    cpu_baseline leg of bench.py and by full-size property tests.
    """
    M, D = testing.shape
    mu = np.empty(M)
    var = np.empty(M)
    deriv = np.empty((M, D))
    for s in range(0, M, block):
        e = min(M, s + block)
        mu[s:e], var[s:e], deriv[s:e] = cpu_predict(inputs, theta, invQ, invQt,
                                                    testing[s:e])
    return mu, var, deriv


def multivariate_predict(basis_functions, emulators, y):
    """MultivariateEmulator.predict for ONE test point.

    gp_emulator/multivariate_gp.py:195-222.  ``emulators`` is a list of
    ``(inputs, theta, invQ, invQt)`` tuples, one per principal component.
    """
    y = np.atleast_2d(y)
    fwd = np.zeros(basis_functions[0].shape[0])
    deriv = np.zeros((y.shape[1], basis_functions.shape[1]))
    for i, (inputs, theta, invQ, invQt) in enumerate(emulators):
        mu, var, grad = cpu_predict(inputs, theta, invQ, invQt, y)
        fwd += mu * basis_functions[i]
        deriv += np.asarray(grad).T @ np.atleast_2d(basis_functions[i])
    return fwd.squeeze(), deriv


def benchmark_inputs(seed, n_train, n_inputs, n_predict):
    """The synthetic recipe of tests/benchmark.py:11-15,28-29, but seeded.

    Order of draws is fixed here and shared by fixtures, tests and bench.py:
    inputs, testing, theta, invQ (deliberately non-symmetric), invQt.
    """
    rs = np.random.RandomState(seed)
    inputs = rs.random_sample((n_train, n_inputs))
    testing = rs.random_sample((n_predict, n_inputs))
    theta = rs.random_sample(n_inputs + 2)
    invQ = rs.random_sample((n_train, n_train))
    invQt = rs.random_sample(n_train)
    return inputs, testing, theta, invQ, invQt


def maxnorm_err(ref, got):
    """The reference's parity metric: max|ref-got| / max|ref| (tests/benchmark.py:51-53)."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return float(np.max(np.abs(ref - got)) / np.max(np.abs(ref)))
