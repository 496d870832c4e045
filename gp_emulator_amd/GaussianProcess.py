"""Host-side mirror of the reference's ``GaussianProcess`` for the predict path.

Same class name, method names, argument meaning and return shapes as
gp_emulator/GaussianProcess.py, so scripts written against the reference
(tests/benchmark.py, tests/test_perband_emulator.py) keep working:

  predict(testing, do_unc=True, is_gpu=False, precision=np.float64, threshold=2e5)  :327-341
  gpu_predict(testing, precision, threshold)                                          :273-323
  get_gpu_block(size, block_size)                                                     :253-270
  cpu_predict(testing, do_unc=True)                                                   :211-251
  hessian(testing)                                                                    :345-366
  _set_params(theta) / _prepare_likelihood()                                          :52-75,127-139

``is_gpu=True`` runs the fused HIP kernel through ``_gpu_predict.predict_wrap`` (the same
twelve-argument boundary the reference crosses at :313-316).  It never falls back to the
CPU: if the library or the GPU is missing it raises ``GpuPredictUnavailable``.
``is_gpu=False`` is the reference API's explicit numpy branch and is only taken when the
caller asks for it.  Training (``learn_hyperparameters``, ``loglikelihood``,
``partial_devs``, :77-125,141-209) is SURVEY.md section 8(f) rank 2, the first thing after
the predict path: the objective and its gradient run on the GPU (``is_gpu=True``,
``csrc/gp_train_kernel.hpp``), the L-BFGS-B driver stays scipy on the host.

Unlike the reference (:7) this module does not import the GPU extension at import time.
"""
import random

import numpy as np
import scipy.spatial.distance as dist

__all__ = ["GaussianProcess", "k_fold_cross_validation"]


def k_fold_cross_validation(X, K, randomise=False):
    """K (training, validation) partitions of the items of ``X`` (reference :9-26): item i goes
    to the validation list of fold ``i % K`` and to the training list of every other fold;
    ``randomise`` shuffles a copy first (``random.shuffle``, the ``random`` module's stream)."""
    if randomise:
        X = list(X)
        random.shuffle(X)
    for k in range(K):
        training = [x for i, x in enumerate(X) if i % K != k]
        validation = [x for i, x in enumerate(X) if i % K == k]
        yield training, validation


class GaussianProcess:
    """Squared-exponential ARD Gaussian-process emulator (predict side)."""

    # True: gpu_predict makes ONE call on a device-resident model (gp_predict_host: constants
    # uploaded once per emulator, rows through the library's slab pipeline, row-major gradient
    # written straight into the array it returns).  False: the reference's own flow, block by
    # block through the twelve-argument predict_wrap with the un-transpose of :313-321.
    row_major_boundary = True

    def __init__(self, inputs, targets):
        # gp_emulator/GaussianProcess.py:35-51
        self.inputs = inputs
        self.targets = targets
        (self.n, self.D) = self.inputs.shape
        self._gpu_models = {}

    # ------------------------------------------------------------------ inputs of the path
    def _prepare_likelihood(self):
        """Q, invQ, invQt, logdetQ from theta (reference :52-75).  Host LAPACK, run once."""
        e = np.exp(self.theta)
        x = np.asarray(self.inputs, dtype=np.float64)
        Z = np.zeros((self.n, self.n))
        for d in range(self.D):
            col = x[:, d]
            Z = Z + e[d] * (col[None, :] - col[:, None]) ** 2
        self.Z = e[self.D] * np.exp(-0.5 * Z)
        self.Q = self.Z + e[self.D + 1] * np.eye(self.n)
        self.invQ = np.linalg.inv(self.Q)
        self.invQt = np.dot(self.invQ, self.targets)
        self.logdetQ = 2.0 * np.sum(np.log(np.diag(np.linalg.cholesky(self.Q))))
        self._gpu_models = {}

    def _set_params(self, theta, is_gpu=False):
        """Set hyper-parameters and precompute what predict needs (reference :127-139).
        ``is_gpu=True`` takes invQ and invQt from the HIP likelihood kernel instead of host
        LAPACK (``Z``, ``Q`` and ``logdetQ``, which only the numpy training branch reads, are
        then not formed)."""
        if is_gpu == True:  # noqa: E712
            self.loglikelihood(theta, is_gpu=True)
            return
        self.theta = theta
        self._prepare_likelihood()

    # ------------------------------------------------------------------ training objective
    def loglikelihood(self, theta, is_gpu=False):
        """The reference's cost (negative log marginal likelihood, reference :77-95):
        0.5 logdetQ + 0.5 t.invQt + 0.5 n log(2 pi).  Like the reference it leaves the
        object set to ``theta`` (invQ, invQt ready for predict).  ``is_gpu=True`` evaluates
        cost, gradient, invQ and invQt in one launch of the HIP likelihood kernel
        (``gp_likelihood_batch_f64``) and keeps the gradient for ``partial_devs``."""
        theta = np.asarray(theta, dtype=np.float64)
        if is_gpu == True:  # noqa: E712
            from . import _lib
            ctx = _lib.default_context()
            cost, grad, invQ, invQt = ctx.likelihood_batch(theta[None, :], self.inputs,
                                                           self.targets, want_inverse=True)
            if not np.isfinite(cost[0]):
                # a pivot of the elimination was <= 0: what numpy's Cholesky reports as
                # LinAlgError on the reference's path (:66, caught in _learn, :176-181)
                raise np.linalg.LinAlgError("Matrix is not positive definite")
            self.theta = theta
            self.invQ, self.invQt = invQ[0], invQt[0]
            self._gpu_models = {}
            self._last_grad = (theta.copy(), grad[0])
            loglikelihood = float(cost[0])
        else:
            self._set_params(theta)
            loglikelihood = (0.5 * self.logdetQ + 0.5 * np.dot(self.targets, self.invQt) +
                             0.5 * self.n * np.log(2. * np.pi))
        self.current_theta = theta
        self.current_loglikelihood = loglikelihood
        return loglikelihood

    def partial_devs(self, theta, is_gpu=False):
        """Gradient of the cost w.r.t. the D+2 hyper-parameters (reference :97-125).  As in the
        reference it is evaluated at the state the last ``loglikelihood`` call left; the GPU
        branch returns the gradient that call already produced (re-evaluating if ``theta``
        differs)."""
        if is_gpu == True:  # noqa: E712
            last = getattr(self, "_last_grad", None)
            if last is None or not np.array_equal(last[0], np.asarray(theta, dtype=np.float64)):
                self.loglikelihood(theta, is_gpu=True)
                last = self._last_grad
            return last[1].copy()
        x = np.asarray(self.inputs, dtype=np.float64)
        partial_d = np.zeros(self.D + 2)
        for d in range(self.D):
            col = x[:, d]
            V = ((col[None, :] - col[:, None]) ** 2) * self.Z
            partial_d[d] = np.exp(self.theta[d]) * (
                np.dot(self.invQt, np.dot(V, self.invQt)) - np.sum(self.invQ * V)) / 4.
        partial_d[self.D] = 0.5 * np.sum(self.invQ * self.Z) - \
            0.5 * np.dot(self.invQt, np.dot(self.Z, self.invQt))
        e_noise = np.exp(self.theta[self.D + 1])
        partial_d[self.D + 1] = 0.5 * np.trace(self.invQ) * e_noise - \
            0.5 * np.dot(self.invQt, self.invQt) * e_noise
        return partial_d

    def _learn(self, theta0, verbose, is_gpu=False):
        """One L-BFGS-B minimisation from ``theta0`` (reference :141-181: factr=0.1,
        pgtol=1e-20; a LinAlgError returns the last point with cost 9999)."""
        import warnings
        from scipy.optimize import fmin_l_bfgs_b
        iprint = 1 if verbose else -1
        try:
            if is_gpu:
                def both(th):
                    f = self.loglikelihood(th, is_gpu=True)
                    return f, self.partial_devs(th, is_gpu=True)
                theta_opt = fmin_l_bfgs_b(both, theta0, factr=0.1, pgtol=1e-20, iprint=iprint)
            else:
                theta_opt = fmin_l_bfgs_b(self.loglikelihood, theta0, fprime=self.partial_devs,
                                          factr=0.1, pgtol=1e-20, iprint=iprint)
        except np.linalg.LinAlgError:
            warnings.warn("Optimisation resulted in linear algebra error. Returning last "
                          "loglikelihood calculated, but this is fishy", RuntimeWarning)
            theta_opt = [self.current_theta, 9999]
        return theta_opt

    def learn_hyperparameters(self, n_tries=15, verbose=False, is_gpu=False):
        """Fit the hyper-parameters from ``n_tries`` random starts ``5 (rand - 0.5)`` and keep
        the best (reference :183-209).  Returns ``(cost, theta)`` and leaves the object set to
        it.  With ``is_gpu=True`` every cost/gradient evaluation is one launch of the HIP
        likelihood kernel."""
        log_like = []
        params = []
        starts = 5. * (np.random.rand(n_tries, self.D + 2) - 0.5)
        if is_gpu and n_tries > 1:
            # one evaluation keeps a single workgroup busy, so the restarts advance together and
            # every round of cost/gradient requests is one batched launch (perband.learn_bands:
            # scipy's L-BFGS-B stepped in reverse communication, same starting points)
            from . import perband
            costs, thetas, _ = perband.learn_bands([self], starts=starts[None, :, :])
            if verbose:
                print("After %d, the minimum cost was %e" % (n_tries, costs[0]))
            self._last_grad = None
            return (costs[0], thetas[0])
        results = [self._learn(theta, verbose, is_gpu=is_gpu) for theta in starts]
        for T in results:
            log_like.append(T[1])
            params.append(T[0])
        log_like = np.array(log_like)
        idx = np.argsort(log_like)[0]
        if verbose:
            print("After %d, the minimum cost was %e" % (n_tries, log_like[idx]))
        self.loglikelihood(params[idx], is_gpu=is_gpu)
        return (log_like[idx], params[idx])

    # ------------------------------------------------------------------ numpy branch
    def cpu_predict(self, testing, do_unc=True):
        """The reference API's numpy branch (reference :211-251); explicit, never a fallback."""
        (nn, D) = testing.shape
        assert D == self.D
        expX = np.exp(self.theta)
        s = np.sqrt(expX[:D])
        a = dist.cdist(s * self.inputs, s * testing, "sqeuclidean")
        a = expX[D] * np.exp(-0.5 * a)
        b = expX[D]
        mu = np.dot(a.T, self.invQt)
        if do_unc:
            var = b - np.sum(a * np.dot(self.invQ, a), axis=0)
        deriv = np.zeros((nn, D))
        for d in range(D):
            aa = self.inputs[:, d].flatten()[None, :] - testing[:, d].flatten()[:, None]
            deriv[:, d] = expX[d] * np.dot((a * aa.T).T, self.invQt)
        if do_unc:
            return mu, var, deriv
        return mu, deriv

    # ------------------------------------------------------------------ GPU branch
    def get_gpu_block(self, size, block_size):
        """Start/end indices of row blocks no wider than ``block_size``; when there is more
        than one block the last two are re-split evenly (reference :253-270, Python-2
        integer division at :265)."""
        size, block_size = int(size), int(block_size)
        ind_start = np.arange(0, size, block_size, dtype=np.int64)
        ind_end = np.append(ind_start[1:], size).astype(np.int64)
        nblocks = len(ind_start)
        if nblocks > 1:
            half = (ind_end[nblocks - 1] - ind_start[nblocks - 2]) // 2
            ind_end[nblocks - 2] = ind_start[nblocks - 2] + half
            ind_start[nblocks - 1] = ind_end[nblocks - 2]
        assert np.all(ind_end - ind_start <= block_size)
        return ind_start, ind_end

    def gpu_predict(self, testing, precision, threshold, out=None):
        """Predict on the GPU in row blocks of at most ``threshold`` rows (reference
        :273-323): constants flattened and cast to ``precision`` once (:289-292), each
        block's rows flattened and cast (:300-311), ``predict_wrap`` called with the twelve
        reference arguments (:313-316), ``deriv`` un-transposed from the boundary's
        dimension-major layout (:321).  Outputs are float64 whatever ``precision`` is, as
        in the reference (it appends onto float64 arrays, :318-321).

        ``threshold`` only bounds the block size; the kernel itself has no 2e5-row limit
        (the reference's came from kernel_matrixExp.cu:29).
        """
        precision = np.dtype(precision)
        if precision not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("precision must be float32 or float64, got %r" % (precision,))
        n_predict, n_inputs = testing.shape
        assert n_inputs == self.D
        if self.row_major_boundary:
            # One library call: the constants are packed and uploaded once per emulator
            # (gpu_model), the rows flow through the library's slab pipeline with at most
            # ``threshold`` rows per launch, and the row-major gradient lands in the array that
            # is returned -- no per-block upload, append or transpose (reference :297-321).
            # float32 on float64 data: the library converts while it stages.
            testing = np.asarray(testing)
            if testing.dtype != np.float64:
                testing = testing.astype(precision)
            model = self.gpu_model(precision)
            # ``out=(result, error, deriv)``: the caller's own arrays; when they AND ``testing`` are page-locked
            # (``gp_emulator_amd.pinned_empty``) the library copies straight between them and the device
            result, error, deriv = model.predict(testing, max_block_rows=int(threshold), out=out)
            if result.dtype != np.float64:   # the reference's outputs are float64 (:318-321)
                result, error, deriv = (a.astype(np.float64) for a in (result, error, deriv))
            return result, error, deriv

        # The reference's own flow, block by block through its twelve-argument boundary.
        from . import _gpu_predict
        n_train = self.inputs.shape[0]
        theta_size = self.theta.size

        def cast(a, size):   # 1-D, contiguous, ``precision`` (reference :289-292)
            return np.ascontiguousarray(np.asarray(a).reshape(size), dtype=precision)
        inputs = cast(self.inputs, n_train * n_inputs)
        invQt = cast(self.invQt, n_train)
        invQ = cast(self.invQ, n_train * n_train)
        expX = cast(np.exp(self.theta), theta_size)

        result = np.empty(n_predict)
        error = np.empty(n_predict)
        deriv = np.empty((n_predict, n_inputs))
        ind_start, ind_end = self.get_gpu_block(n_predict, threshold)
        for block_start, block_end in zip(ind_start, ind_end):
            n_blk = int(block_end - block_start)
            testing_block = cast(testing[block_start:block_end, :], n_blk * n_inputs)
            result_block = np.zeros(n_blk, dtype=precision)
            error_block = np.zeros(n_blk, dtype=precision)
            deriv_block = np.zeros(n_blk * n_inputs, dtype=precision)
            _gpu_predict.predict_wrap(expX, inputs, invQt, invQ, testing_block,
                                      result_block, error_block, deriv_block,
                                      n_blk, n_train, n_inputs, theta_size)
            deriv[block_start:block_end, :] = deriv_block.reshape(n_inputs, n_blk).T
            result[block_start:block_end] = result_block
            error[block_start:block_end] = error_block
        return result, error, deriv

    def predict(self, testing, do_unc=True, is_gpu=False, precision=np.float64, threshold=2e5, out=None):
        """Mean, variance and gradient at ``testing`` (n_predict, n_inputs) (reference
        :327-341).  ``do_unc`` only affects the numpy branch, as in the reference.  ``out`` (GPU branch only, not in
        the reference): three arrays to fill instead of pooled ones, e.g. page-locked ones (``pinned_empty``)."""
        if is_gpu == True:  # noqa: E712  (reference spelling, :338)
            return self.gpu_predict(testing, precision, threshold=threshold, out=out)
        return self.cpu_predict(testing, do_unc)

    # ------------------------------------------------------------------ device-resident use
    def gpu_model(self, precision=np.float64, device=None):
        """Constants packed and uploaded once (cached per precision/device): what ``predict``
        and ``hessian`` run on, and the form to use when the test rows already live in HBM.
        The cache entry remembers the constants it was built from and is rebuilt when theta,
        inputs, invQt or invQ no longer compare equal (they are plain attributes, as in the
        reference, and callers assign them)."""
        from . import _lib
        device = _lib.default_device() if device is None else int(device)
        key = (np.dtype(precision).str, device)
        invQ = getattr(self, "invQ", None)        # absent: a model for the Hessian alone
        consts = (np.asarray(self.theta), np.asarray(self.inputs), np.asarray(self.invQt),
                  np.asarray(invQ) if invQ is not None else np.empty(0))
        hit = self._gpu_models.get(key)
        if hit is not None:
            m, kept = hit
            if all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(consts, kept)):
                return m
            m.close()
        ctx = _lib.default_context(device)
        m = _lib.Model(ctx, np.exp(self.theta), self.inputs, self.invQt, invQ, precision)
        self._gpu_models[key] = (m, tuple(np.array(a, copy=True) for a in consts))
        return m

    # ------------------------------------------------------------------ Hessian
    def hessian(self, testing, is_gpu=False, precision=np.float64):
        """(nn, D, D) Hessian of the mean (reference :345-366, which is numpy only and takes
        just ``testing``).  ``is_gpu=True`` runs the fused HIP kernel through the C ABI
        (``gp_hessian_host`` on the cached device model) and, like ``predict``, never falls back to the CPU.
        ``is_gpu=False`` is the explicit numpy branch, written as the symmetric rank-N update
        it is: H = sum_i w_i u_i u_i^T - diag(e) mu,  w = a * invQt,  u_i = e * (x_i - t)."""
        (nn, D) = testing.shape
        assert D == self.D
        if is_gpu == True:  # noqa: E712
            dt = np.dtype(precision)
            if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
                raise TypeError("precision must be float32 or float64")
            hess = self.gpu_model(dt).hessian(np.asarray(testing))
            return hess.astype(np.float64, copy=False)
        expX = np.exp(self.theta)
        s = np.sqrt(expX[:D])
        a = expX[D] * np.exp(-0.5 * dist.cdist(s * self.inputs, s * testing, "sqeuclidean"))
        w = a * np.asarray(self.invQt)[:, None]                      # (n, nn)
        hess = np.zeros((nn, D, D))
        u = expX[:D][None, None, :] * (np.asarray(self.inputs)[:, None, :] - testing[None, :, :])
        for d in range(D):
            for d2 in range(D):
                hess[:, d, d2] = np.sum(w * u[:, :, d] * u[:, :, d2], axis=0)
            hess[:, d, d] -= expX[d] * np.sum(w, axis=0)
        return hess
