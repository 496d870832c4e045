"""Host-side mirror of the reference's ``MultivariateEmulator`` (predict side).

gp_emulator/multivariate_gp.py:38-222: a multivariate model output (e.g. a 2101-band
spectrum) is compressed onto ``n_pcs`` principal components and each PC weight is emulated
by its own ``GaussianProcess`` on the shared training parameters ``y``.  This mirror keeps
the constructor, ``compress``, ``predict`` and ``dump_emulator`` with the reference's
argument names and return shapes, and loads the reference's ``.npz`` dumps (:68-79,
e.g. data/prosail_30_0_30_0.npz) with ``allow_pickle=False``.

With ``hyperparams`` (given or from ``dump``) every per-PC emulator is only set up
(``_set_params``, reference :185-186); without, each one is trained with
``learn_hyperparameters(n_tries)`` (:180-184).  ``is_gpu=True`` -- a keyword the reference's
constructor does not have -- does either on the GPU: all n_pcs inverses in ONE launch of the
likelihood kernel (they share the training inputs), or the GPU training objective.
"""
import numpy as np

from .GaussianProcess import GaussianProcess

__all__ = ["MultivariateEmulator"]


class MultivariateEmulator(object):

    def __init__(self, dump=None, X=None, y=None, hyperparams=None, thresh=0.98, n_tries=5,
                 basis_functions=None, n_pcs=None, is_gpu=False):
        # reference :40-122; basis_functions / n_pcs: a stored decomposition (what
        # save_emulators.py:93-98 tries to pass), is_gpu: set up or train on the GPU
        if dump is not None:
            if X is None and y is None:
                with np.load(dump, allow_pickle=False) as f:
                    X = f["X"]
                    y = f["y"]
                    hyperparams = f["hyperparams"]
                    thresh = float(f["thresh"])
                    if "basis_functions" in f.files:
                        basis_functions = f["basis_functions"]
                        n_pcs = int(f["n_pcs"])
            else:
                raise ValueError("You specified both a dump file and X and y")
        else:
            if X is None or y is None:
                raise ValueError("Need to specify both X and y")
            assert X.shape[0] == y.shape[0]
            assert X.ndim == 2
            assert y.ndim == 2

        self.X_train = X
        self.y_train = y
        self.thresh = thresh
        if basis_functions is None:
            self.calculate_decomposition(X, thresh)
            basis_functions = self.basis_functions
            n_pcs = self.n_pcs
        self.n_pcs = int(n_pcs)
        self.basis_functions = basis_functions
        if hyperparams is not None:
            assert (y.shape[1] + 2 == hyperparams.shape[0]) and (self.n_pcs == hyperparams.shape[1])
        self.train_emulators(X, y, hyperparams=hyperparams, n_tries=n_tries, is_gpu=is_gpu)

    def dump_emulator(self, fname):
        """Save in the reference's .npz layout (reference :124-137)."""
        np.savez_compressed(fname, X=self.X_train, y=self.y_train, hyperparams=self.hyperparams,
                            thresh=self.thresh, basis_functions=self.basis_functions,
                            n_pcs=self.n_pcs)

    def calculate_decomposition(self, X, thresh):
        """PCA by SVD; keep the PCs whose cumulative singular-value share is <= thresh
        (reference :139-160)."""
        U, s, V = np.linalg.svd(X, full_matrices=True)
        pcnt_var_explained = s.cumsum() / s.sum()
        # the reference indexes V (N_full rows) with a mask of len(s) = min(N_train, N_full)
        # entries (:158), which old numpy tolerated; select among the first len(s) rows
        self.basis_functions = V[:s.size][pcnt_var_explained <= thresh]
        self.n_pcs = int(np.sum(pcnt_var_explained <= thresh))

    def train_emulators(self, X, y, hyperparams, n_tries=2, is_gpu=False):
        """One GaussianProcess per PC on the shared inputs (reference :162-188)."""
        self.emulators = []
        train_data = self.compress(X)
        self.hyperparams = np.zeros((2 + y.shape[1], self.n_pcs))
        for i in range(self.n_pcs):
            self.emulators.append(GaussianProcess(np.atleast_2d(y), train_data[i]))
        if hyperparams is None:
            if is_gpu and self.n_pcs > 1:
                # all PCs and restarts together (same starting points as the loop below would draw)
                from . import perband
                costs, thetas, _ = perband.learn_bands(self.emulators, n_tries=n_tries)
                self.hyperparams[:, :] = thetas.T
                return
            for i, gp in enumerate(self.emulators):
                self.hyperparams[:, i] = gp.learn_hyperparameters(n_tries=n_tries, is_gpu=is_gpu)[1]
            return
        self.hyperparams[:, :] = np.asarray(hyperparams)[:, :self.n_pcs]
        if is_gpu and self.n_pcs > 0:
            # the emulators differ in targets and theta only: one launch, one workgroup each
            from . import _lib
            cost, grad, invQ, invQt = _lib.default_context().likelihood_batch(
                self.hyperparams.T, np.atleast_2d(y), train_data, want_inverse=True)
            bad = np.flatnonzero(~np.isfinite(cost))
            if bad.size:
                # a pivot of the elimination was <= 0: what the host branch (_set_params ->
                # numpy's Cholesky, reference :66) reports as LinAlgError
                raise np.linalg.LinAlgError("Matrix is not positive definite (principal component %d)"
                                            % int(bad[0]))
            for i, gp in enumerate(self.emulators):
                gp.theta = self.hyperparams[:, i].copy()
                gp.invQ, gp.invQt = invQ[i], invQt[i]
                gp.current_theta, gp.current_loglikelihood = gp.theta, float(cost[i])
            return
        for i, gp in enumerate(self.emulators):
            gp._set_params(hyperparams[:, i])

    def compress(self, X):
        """Project full-rank vectors onto the PC basis (reference :190-193)."""
        return X.dot(self.basis_functions.T).T

    def predict(self, y, do_deriv=True, is_gpu=False):
        """Reconstructed output and its Jacobian at ONE input vector ``y`` (reference
        :195-222): ``fwd (N_full,)`` and ``deriv (N_params, N_full)``.  The numpy branch is the
        reference's loop over the per-PC emulators.  ``is_gpu=True`` -- the call an optimiser makes
        once per state vector -- runs on the device-resident form of the whole emulator
        (``predict_many`` on one row: all PCs in one launch, reconstruction and Jacobian on the
        device) instead of handing the flag to each per-PC ``predict`` as the reference does
        (:215): same numbers, one round trip instead of ``n_pcs``."""
        y = np.atleast_2d(y)
        if is_gpu:
            if y.shape[0] != 1:
                raise ValueError("predict takes one input vector; predict_many takes rows")
            if do_deriv:
                fwd, jac = self.predict_many(y, is_gpu=True, do_deriv=True)
                return fwd[0], jac[0]
            return self.predict_many(y, is_gpu=True)[0]
        fwd = np.zeros(self.basis_functions[0].shape[0])
        if do_deriv:
            deriv = np.zeros((y.shape[1], self.basis_functions.shape[1]))
        for i in range(self.n_pcs):
            pred_mu, pred_var, grad = self.emulators[i].predict(y)
            fwd += pred_mu * self.basis_functions[i]
            if do_deriv:
                deriv += np.asarray(grad).T @ np.atleast_2d(self.basis_functions[i])
        if do_deriv:
            return fwd.squeeze(), deriv
        return fwd.squeeze()

    # ---- the emulator resident on the device ----------------------------------------------------
    def _host_arrays(self):
        """Everything the device-resident copy is made from: every emulator's constants and the basis."""
        out = []
        for gp in self.emulators:
            out += [gp.theta, gp.inputs, gp.invQt, gp.invQ]
        out.append(self.basis_functions)
        return out

    def _gpu_state(self, dt):
        """The device-resident form of the whole emulator (packed per-PC emulators + basis), rebuilt whenever
        the host data is no longer what it was made from.  The reference uploads every constant on every call
        (gpu/predict.cu:11-34), so a caller may replace OR edit in place theta, invQ, invQt, inputs or the basis
        between two calls and the next call computes with the new values; so it is here: the state remembers the
        array objects (cheap pre-filter) and a 64-bit digest of all their bytes (``_lib.HostBlocks``), and the
        digest is re-taken on EVERY call -- inside the library call, while the device works and the calling
        thread would only wait (``gp_mv_predict_host_checked``), so the latency path pays nothing for it; a
        mismatch discards that call's results, rebuilds the state and calls again."""
        from . import _lib, perband
        cache = self.__dict__.setdefault("_gpu", {})
        key = (dt.str, _lib.default_device())
        arrays = self._host_arrays()
        st = cache.get(key)
        if st is not None and st["blocks"].same_arrays(arrays):
            return st
        if st is not None:
            self._release(st)
            del cache[key]
        blocks = _lib.HostBlocks(arrays)           # (digest first: what the batch below is packed from)
        batch = perband.make_batch(self.emulators, dt)
        ctx = batch.ctx
        st = {"blocks": blocks, "batch": batch, "ctx": ctx,
              "d_basis": ctx.to_device(np.ascontiguousarray(self.basis_functions, dtype=dt))}
        cache[key] = st
        return st

    @staticmethod
    def _release(st):
        st["ctx"].free(st["d_basis"])
        st["batch"].close()

    def release_gpu(self):
        """Free the device-resident copy (packed emulators, basis); the next
        ``is_gpu=True`` call rebuilds it."""
        for st in self.__dict__.get("_gpu", {}).values():
            self._release(st)
        self.__dict__["_gpu"] = {}

    def __del__(self):
        try:
            self.release_gpu()
        except Exception:
            pass

    def predict_many(self, Y, is_gpu=True, precision=np.float64, do_deriv=False):
        """Beyond the reference (whose predict breaks for more than one row, SURVEY.md
        section 3.3): reconstructed outputs ``(M, N_full)`` -- and with ``do_deriv`` the
        Jacobians ``(M, N_params, N_full)`` -- for M input rows.  On the GPU the n_pcs
        emulators run as ONE batched launch over the shared rows and the reconstruction
        ``sum_pc mu_pc * basis_pc`` is a second kernel on the resident outputs; only the
        reconstructed arrays cross PCIe.  The packed emulators and the basis stay on the device
        between calls (``release_gpu()`` frees them), and a call is one library call
        (``gp_mv_predict_host``): rows up, three launches, results down."""
        Y = np.atleast_2d(Y)
        M, D = Y.shape
        B = self.basis_functions.shape[1]
        if not is_gpu:
            out = [gp.predict(Y) for gp in self.emulators]
            fwd = np.stack([o[0] for o in out]).T @ self.basis_functions
            if not do_deriv:
                return fwd
            grads = np.stack([o[2] for o in out])                  # (P, M, D)
            return fwd, np.einsum("pmd,pb->mdb", grads, self.basis_functions)
        from . import _lib
        dt = np.dtype(precision)
        st = self._gpu_state(dt)
        ctx = st["ctx"]
        isz = dt.itemsize
        Yc = np.ascontiguousarray(Y, dtype=dt)
        # one library call per <= 1 GiB of results (rows up, three launches, results down, one
        # synchronisation): a single call for anything an optimiser asks for
        step = max(1, (1 << 30) // (B * (1 + (D if do_deriv else 0)) * isz))
        if do_deriv and M <= step:                         # one call: fwd and jac back to back, one copy down
            both = ctx.out_pool.take((M * B * (1 + D),), dt)   # >= 1 MB: recycled memory (see _lib.OutputPool)
            fwd, jac = both[:M * B].reshape(M, B), both[M * B:].reshape(M, D, B)
        else:
            fwd = ctx.out_pool.take((M, B), dt)
            jac = ctx.out_pool.take((M, D, B), dt) if do_deriv else None
        r0 = 0
        while r0 < M:
            r1 = min(M, r0 + step)
            blocks = st["blocks"].refresh()
            rc = ctx.lib.gp_mv_predict_host_checked(ctx.h, st["batch"].h, st["d_basis"], _lib._ptr(Yc[r0:r1]), r1 - r0, B,
                                                    _lib._ptr(fwd[r0:r1]), _lib._ptr(jac[r0:r1]) if do_deriv else None,
                                                    blocks.ptrs, blocks.lens, blocks.n, blocks.expected)
            if rc == _lib.GP_STALE:
                # the host arrays were edited in place since the resident copy was made: rebuild it from what
                # they hold now and run the same rows again (nothing of the stale call is kept)
                self._release(st)
                del self.__dict__["_gpu"][(dt.str, _lib.default_device())]
                st = self._gpu_state(dt)
                ctx = st["ctx"]
                continue
            _lib.check(rc, "gp_mv_predict_host_checked")
            r0 = r1
        return (fwd, jac) if do_deriv else fwd
