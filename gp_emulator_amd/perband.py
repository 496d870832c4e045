"""Per-band emulators predicted together.

The reference's per-band pattern (tests/test_perband_emulator.py:22-37) builds one
``GaussianProcess`` per spectral band on the SAME training inputs ``y_train`` -- each with
its own targets, theta, invQ, invQt -- and then calls ``gp.predict(X, is_gpu=True)`` band by
band (:48-50).  ``predict_bands`` runs all of them over the shared test rows in one launch
of the fused kernel (BASELINE config 3: 2101 bands x N_train=250 x D=11, shared test set).
"""
import numpy as np

from . import _lib


def make_batch(gps, precision=np.float64, device=None):
    """Pack a list of GaussianProcess objects sharing ``inputs`` into one BatchModel."""
    if not gps:
        raise ValueError("need at least one GaussianProcess")
    inputs = np.asarray(gps[0].inputs)
    for gp in gps[1:]:
        other = np.asarray(gp.inputs)
        if other.shape != inputs.shape or not np.array_equal(other, inputs):
            raise ValueError("per-band emulators must share the same training inputs")
    expX = np.stack([np.exp(gp.theta) for gp in gps])
    invQt = np.stack([np.asarray(gp.invQt) for gp in gps])
    invQ = np.stack([np.asarray(gp.invQ) for gp in gps])
    return _lib.BatchModel(_lib.default_context(device), expX, inputs, invQt, invQ, precision)


def predict_bands(gps, testing, precision=np.float64, device=None):
    """mu (E, M), var (E, M), deriv (E, M, D) for the E emulators in ``gps``."""
    batch = make_batch(gps, precision, device)
    try:
        return batch.predict(np.asarray(testing))
    finally:
        batch.close()


# ---------------------------------------------------------------------------------------------
# Training many per-band emulators at once
# ---------------------------------------------------------------------------------------------
class _GatheredObjective(object):
    """Turns many concurrent single-theta evaluations into few batched ones.

    Each optimiser thread calls ``evaluate(slot, theta)`` and blocks.  When every thread that
    is still optimising has a request in, the thread that completed the set runs ONE batched
    evaluation for all of them (``batch_fn(thetas, targets) -> (cost, grad)``; on the GPU one
    launch of the likelihood kernels, one workgroup per request) and wakes the others.  No
    dispatcher thread, no polling; the scipy L-BFGS-B instances stay ordinary and independent.

    Synchronisation is deliberately primitive: one short critical section to register a request,
    and one private lock per thread to sleep on (released by whoever ran the batch).  A shared
    ``Condition`` + ``notify_all`` makes every woken thread queue for the same lock again and,
    with a few hundred threads, costs several times more than the evaluations themselves.
    """

    def __init__(self, batch_fn, n_slots, n_train, n_theta):
        import threading
        self._fn = batch_fn
        self._lock = threading.Lock()
        self._wake = [threading.Lock() for _ in range(n_slots)]
        for w in self._wake:
            w.acquire()                 # held: a waiter blocks until the batch runner releases it
        self._thetas = np.zeros((n_slots, n_theta))
        self._targets = np.zeros((n_slots, n_train))
        self._cost = np.zeros(n_slots)
        self._grad = np.zeros((n_slots, n_theta))
        self._waiting = []              # slots with a request in
        self._active = n_slots          # threads that may still submit requests
        self._error = None
        self.launches = 0
        self.evaluations = 0

    def set_targets(self, slot, targets):
        self._targets[slot] = targets

    def _run(self, batch, me):
        # every live thread is in `batch` and asleep (or is `me`): nothing else touches the state
        idx = np.array(batch)
        try:
            cost, grad = self._fn(self._thetas[idx], self._targets[idx])
            self._cost[idx] = cost
            self._grad[idx] = grad
        except BaseException as exc:    # wake everybody, re-raise in every thread
            self._error = exc
        self.launches += 1
        self.evaluations += len(batch)
        for s in batch:
            if s != me:
                self._wake[s].release()

    def evaluate(self, slot, theta):
        self._thetas[slot] = theta
        with self._lock:
            self._waiting.append(slot)
            batch = None
            if len(self._waiting) >= self._active:
                batch, self._waiting = self._waiting, []
        if batch is not None:
            self._run(batch, slot)
        else:
            self._wake[slot].acquire()
        if self._error is not None:
            raise self._error
        return float(self._cost[slot]), self._grad[slot].copy()

    def retire(self):
        """The calling thread will submit no more requests."""
        with self._lock:
            self._active -= 1
            batch = None
            if self._active > 0 and len(self._waiting) >= self._active:
                batch, self._waiting = self._waiting, []
        if batch is not None:
            self._run(batch, -1)


def _lockstep_driver():
    """scipy's reverse-communication L-BFGS-B entry point (``scipy.optimize._lbfgsb.setulb``, the
    routine ``fmin_l_bfgs_b`` itself loops over), or None when this scipy does not have the
    signature written for here (1.15: C translation, integer task codes)."""
    try:
        import inspect
        import scipy
        from scipy.optimize import _lbfgsb, _lbfgsb_py
        major, minor = (int(v) for v in scipy.__version__.split(".")[:2])
        if (major, minor) < (1, 15):
            return None
        if "ln_task" not in inspect.getsource(_lbfgsb_py._minimize_lbfgsb):
            return None
        return _lbfgsb.setulb
    except Exception:
        return None


class _Instance(object):
    """One L-BFGS-B minimisation in reverse communication: the state ``fmin_l_bfgs_b`` keeps in
    local variables (scipy/optimize/_lbfgsb_py.py, ``_minimize_lbfgsb``), same sizes and defaults
    (m = 10, maxls = 20, maxfun = maxiter = 15000, no bounds)."""
    __slots__ = ("e", "k", "x", "f", "g", "wa", "iwa", "task", "ln_task", "lsave", "isave", "dsave",
                 "nit", "nfev", "low", "up", "nbd", "last_good", "failed")

    def __init__(self, e, k, x0):
        n, m = x0.size, 10
        self.e, self.k = e, k
        self.x = np.array(x0, dtype=np.float64)
        self.f = 0.0
        self.g = np.zeros(n)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29)
        self.low, self.up = np.zeros(n), np.zeros(n)
        self.nbd = np.zeros(n, dtype=np.int32)
        self.nit = self.nfev = 0
        self.last_good = self.x.copy()
        self.failed = False


def _learn_lockstep(setulb, batch_fn, targets, starts, max_batch=4096):
    """Every (emulator, restart) optimisation advances through scipy's own L-BFGS-B routine, one
    thread, no callbacks: each instance is stepped until it asks for a cost/gradient, then ALL the
    requests are evaluated in one batched call.  Iterates are those of ``fmin_l_bfgs_b(factr=0.1,
    pgtol=1e-20)`` (reference GaussianProcess.py:168-170) because it is the same routine fed the
    same numbers."""
    E, n_tries, n = starts.shape
    factr, pgtol, m, maxls, maxfun, maxiter = 0.1, 1e-20, 10, 20, 15000, 15000
    live = [_Instance(e, k, starts[e, k]) for e in range(E) for k in range(n_tries)]
    cost = np.full((E, n_tries), np.inf)
    theta = np.zeros((E, n_tries, n))
    launches = evaluations = 0
    while live:
        asking, still = [], []
        for it in live:
            while True:
                setulb(m, it.x, it.low, it.up, it.nbd, it.f, it.g, factr, pgtol, it.wa, it.iwa,
                       it.task, it.lsave, it.isave, it.dsave, maxls, it.ln_task)
                if it.task[0] == 3:                      # wants f and g at x
                    if it.nfev and np.array_equal(it.x, it.last_good):
                        continue                         # same point again: f, g are current
                    asking.append(it)                    # (what scipy's ScalarFunction cache does)
                    break
                if it.task[0] == 1:                      # new iteration
                    it.nit += 1
                    if it.nit >= maxiter:
                        it.task[0], it.task[1] = 5, 504
                    elif it.nfev > maxfun:
                        it.task[0], it.task[1] = 5, 502
                    continue
                cost[it.e, it.k], theta[it.e, it.k] = it.f, it.x   # converged / stopped
                break
        for s0 in range(0, len(asking), max_batch):
            part = asking[s0:s0 + max_batch]
            th = np.stack([it.x for it in part])
            tg = targets[[it.e for it in part]]
            c, g = batch_fn(th, tg)
            launches += 1
            evaluations += len(part)
            for j, it in enumerate(part):
                it.nfev += 1
                if not np.isfinite(c[j]):
                    # a pivot <= 0: the reference's numpy path raises LinAlgError here and its
                    # optimiser loop gives the restart up with cost 9999 (:176-181)
                    cost[it.e, it.k], theta[it.e, it.k] = 9999.0, it.last_good
                    it.failed = True
                else:
                    it.f = float(c[j])
                    it.g = np.ascontiguousarray(g[j], dtype=np.float64)
                    it.last_good = it.x.copy()
        live = [it for it in asking if not it.failed]
    return cost, theta, launches, evaluations


def _leave_set(gps, thetas, inputs, ctx):
    """Leave every emulator set to its optimum (theta, invQ, invQt), as the reference's
    learn_hyperparameters does (:207-209); on the GPU a few launches of 128 inverses each."""
    if ctx is None:
        for e, gp in enumerate(gps):
            gp._set_params(thetas[e].copy())
        return
    step = 128
    for s in range(0, len(gps), step):
        part = gps[s:s + step]
        tg = np.stack([np.asarray(gp.targets, dtype=np.float64) for gp in part])
        c, g, invQ, invQt = ctx.likelihood_batch(thetas[s:s + len(part)], inputs, tg, want_inverse=True)
        for i, gp in enumerate(part):
            if not np.isfinite(c[i]):     # every restart failed: the reference's final
                raise np.linalg.LinAlgError("Matrix is not positive definite")   # loglikelihood raises too
            gp.theta = thetas[s + i].copy()
            gp.invQ, gp.invQt = invQ[i].copy(), invQt[i].copy()
            gp.current_theta, gp.current_loglikelihood = gp.theta, float(c[i])
            gp._gpu_models = {}


def _finite(cost_grad):
    """A non-finite cost means a pivot of the elimination was <= 0 -- what the reference's numpy
    path reports as LinAlgError (GaussianProcess.py:66) and its optimiser loop catches (:176-181)."""
    if not np.isfinite(cost_grad[0]):
        raise np.linalg.LinAlgError("Matrix is not positive definite")
    return cost_grad


def learn_bands(gps, n_tries=5, concurrency=256, is_gpu=True, starts=None, batch_fn=None,
                device=None, verbose=False, method="auto"):
    """``learn_hyperparameters(n_tries)`` for every GaussianProcess in ``gps`` (per-band
    emulators on the SAME training inputs, tests/test_perband_emulator.py:22-37), with the
    optimisations of all bands and all restarts advancing side by side: ``concurrency``
    scipy L-BFGS-B instances run in threads and their cost/gradient requests are gathered
    into batched launches of the likelihood kernels (one workgroup per request).

    Starting points are drawn exactly as a loop of ``gp.learn_hyperparameters(n_tries)`` calls
    would draw them (``5 (rand(n_tries, D+2) - 0.5)`` per emulator, in order, from
    ``numpy.random``), unless ``starts`` (E, n_tries, D+2) is given.  On return every emulator
    is set to its best theta (``theta``, ``invQ``, ``invQt``), as the reference leaves it
    (GaussianProcess.py:183-209).  Returns ``(costs (E,), thetas (E, D+2), stats)``.

    ``batch_fn(thetas, targets) -> (cost, grad)`` replaces the GPU objective (tests use a numpy
    one); ``is_gpu`` must be true otherwise -- there is no silent CPU path.

    ``method``: "lockstep" steps scipy's reverse-communication L-BFGS-B routine directly for every
    (emulator, restart) from one thread and evaluates all pending requests per round in one
    launch -- the same iterates as ``fmin_l_bfgs_b`` at a fraction of its per-call Python cost;
    "threads" runs ``concurrency`` ordinary ``fmin_l_bfgs_b`` calls in threads and gathers their
    requests; "auto" takes lockstep when this scipy exposes the routine with the known signature.
    """
    import threading
    import warnings
    from queue import Queue, Empty
    from scipy.optimize import fmin_l_bfgs_b
    if not gps:
        raise ValueError("need at least one GaussianProcess")
    inputs = np.ascontiguousarray(gps[0].inputs, dtype=np.float64)
    for gp in gps[1:]:
        if np.asarray(gp.inputs).shape != inputs.shape or not np.array_equal(gp.inputs, inputs):
            raise ValueError("per-band emulators must share the same training inputs")
    E, (N, D) = len(gps), inputs.shape
    ctx = None
    if batch_fn is None:
        if not is_gpu:
            raise ValueError("learn_bands runs the objective on the GPU; pass is_gpu=True")
        ctx = _lib.default_context(device)

        def batch_fn(thetas, targets):
            return ctx.likelihood_batch(thetas, inputs, targets)
    if starts is None:
        starts = np.stack([5. * (np.random.rand(n_tries, D + 2) - 0.5) for _ in range(E)])
    starts = np.asarray(starts, dtype=np.float64).reshape(E, -1, D + 2)
    n_tries = starts.shape[1]
    setulb = _lockstep_driver() if method in ("auto", "lockstep") else None
    if method == "lockstep" and setulb is None:
        raise RuntimeError("this scipy does not expose the L-BFGS-B routine learn_bands knows how to drive")
    if method not in ("auto", "lockstep", "threads"):
        raise ValueError("method must be 'auto', 'lockstep' or 'threads'")
    if setulb is not None:
        tg_all = np.stack([np.asarray(gp.targets, dtype=np.float64) for gp in gps])
        best_cost, best_theta, launches, evaluations = _learn_lockstep(setulb, batch_fn, tg_all, starts)
        pick = np.argmin(best_cost, axis=1)
        costs = best_cost[np.arange(E), pick]
        thetas = best_theta[np.arange(E), pick]
        _leave_set(gps, thetas, inputs, ctx)
        stats = {"launches": launches, "evaluations": evaluations, "threads": 1, "method": "lockstep"}
        if verbose:
            print("learn_bands: %d emulators x %d starts, %d evaluations in %d launches" % (
                E, n_tries, evaluations, launches))
        return costs, thetas, stats
    jobs = Queue()
    for e in range(E):
        for k in range(n_tries):
            jobs.put((e, k))
    n_threads = int(max(1, min(concurrency, E * n_tries)))
    gather = _GatheredObjective(batch_fn, n_threads, N, D + 2)
    best_cost = np.full((E, n_tries), np.inf)
    best_theta = np.zeros((E, n_tries, D + 2))
    failures = []

    def worker(slot):
        try:
            while True:
                try:
                    e, k = jobs.get_nowait()
                except Empty:
                    return
                gather.set_targets(slot, np.asarray(gps[e].targets, dtype=np.float64))
                try:
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        th, f, _ = fmin_l_bfgs_b(lambda t: _finite(gather.evaluate(slot, t)),
                                                 starts[e, k], factr=0.1, pgtol=1e-20, iprint=-1)
                    best_cost[e, k], best_theta[e, k] = f, th
                except np.linalg.LinAlgError:      # reference :176-181: keep going, cost 9999
                    best_cost[e, k], best_theta[e, k] = 9999.0, starts[e, k]
        except BaseException as exc:
            failures.append(exc)
        finally:
            gather.retire()

    threads = [threading.Thread(target=worker, args=(s,), daemon=True) for s in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if failures:
        raise failures[0]
    pick = np.argmin(best_cost, axis=1)
    costs = best_cost[np.arange(E), pick]
    thetas = best_theta[np.arange(E), pick]
    _leave_set(gps, thetas, inputs, ctx)
    stats = {"launches": gather.launches, "evaluations": gather.evaluations, "threads": n_threads,
             "method": "threads"}
    if verbose:
        print("learn_bands: %d emulators x %d starts, %d evaluations in %d launches" % (
            E, n_tries, gather.evaluations, gather.launches))
    return costs, thetas, stats
