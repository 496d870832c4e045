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


def predict_bands(gps, testing, precision=np.float64, device=None, devices=None, predict_fn=None):
    """mu (E, M), var (E, M), deriv (E, M, D) for the E emulators in ``gps``.

    ``devices`` (a list of device ids) shards the EMULATORS over the GPUs of a node -- BASELINE
    config 3's other partition: GPU g takes a contiguous block of the E
    bands with the shared test rows replicated (8.8 MB for 1e5 rows) instead of every GPU holding
    all E inverses (1 GB for 2101 bands).  One Python thread and one context per device (the C
    ABI releases the GIL), every device's slab pipeline writing its own ``[e0:e1]`` slice of the
    three output arrays: a host gather, no collective.  ``predict_fn(device, gps_block, testing)
    -> (mu, var, deriv)`` replaces the HIP path in the CPU tests of the sharding logic."""
    testing = np.asarray(testing)
    if devices is None:
        batch = make_batch(gps, precision, device)
        try:
            return batch.predict(testing)
        finally:
            batch.close()
    import threading
    from . import multi_gpu
    if not gps:
        raise ValueError("need at least one GaussianProcess")
    E, (M, D) = len(gps), testing.shape
    dt = np.float64 if (testing.dtype == np.float64 or np.dtype(precision) == np.float64) else np.float32
    mu, var, deriv = np.empty((E, M), dt), np.empty((E, M), dt), np.empty((E, M, D), dt)
    blocks = multi_gpu.row_shards(E, len(devices))       # contiguous blocks of emulators
    errors = []

    def work(dev, e0, e1):
        try:
            if e1 <= e0:
                return
            if predict_fn is not None:
                mu[e0:e1], var[e0:e1], deriv[e0:e1] = predict_fn(dev, gps[e0:e1], testing)
                return
            ctx, lock = multi_gpu._device_context(dev)
            with lock:
                part = gps[e0:e1]
                batch = _lib.BatchModel(ctx, np.stack([np.exp(gp.theta) for gp in part]), np.asarray(part[0].inputs),
                                        np.stack([np.asarray(gp.invQt) for gp in part]),
                                        np.stack([np.asarray(gp.invQ) for gp in part]), precision)
                try:
                    batch.predict(testing, out=(mu[e0:e1], var[e0:e1], deriv[e0:e1]))
                finally:
                    batch.close()
        except BaseException as exc:          # surfaced to the caller below
            errors.append(exc)

    inputs = np.asarray(gps[0].inputs)
    for gp in gps[1:]:
        if np.asarray(gp.inputs).shape != inputs.shape or not np.array_equal(gp.inputs, inputs):
            raise ValueError("per-band emulators must share the same training inputs")
    threads = [threading.Thread(target=work, args=(dev, e0, e1)) for dev, (e0, e1) in zip(devices, blocks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return mu, var, deriv


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


_driver_probe = []          # [setulb or None] once probed in this process


def _probe_setulb(setulb):
    """True when ``setulb`` behaves like the routine _Population is written for: called with that
    argument list on f(x) = (x0 - 1)^2 + (x1 + 2)^2 it asks for f, g with task 3, reports new
    iterations with task 1 and stops at the minimum within a few evaluations.  A private entry
    point has no compatibility promise: a scipy that renames an argument, changes a work-array size
    or the task codes fails here (an exception or a wrong answer), never inside a training run."""
    pop = _Population(setulb, [(0, 0)], np.array([[3.0, 4.0]]))
    results = None
    for _ in range(60):
        th, es, finished = pop.advance(results)
        if finished:
            e, k, cost, theta = finished[0]
            return bool(np.isfinite(cost) and cost < 1e-12 and np.allclose(theta, [1.0, -2.0], atol=1e-6))
        if th.shape != (1, 2):
            return False
        d = th - np.array([1.0, -2.0])
        results = (np.sum(d * d, axis=1), 2.0 * d)
    return False


def _lockstep_driver(warn=True):
    """scipy's reverse-communication L-BFGS-B entry point (``scipy.optimize._lbfgsb.setulb``, the
    routine ``fmin_l_bfgs_b`` itself loops over), or None -- with a warning, once -- when this scipy
    does not have it in the form written for here (1.15: C translation, integer task codes).  The
    binding is private, so it is PROBED, not trusted: version, then a run on a two-variable
    quadratic (``_probe_setulb``).  ``learn_bands(method="auto")`` then uses the threaded driver
    (one public ``fmin_l_bfgs_b`` per start, objective calls coalesced into batched launches)."""
    if _driver_probe:
        return _driver_probe[0]
    found, why = None, ""
    try:
        import scipy
        from scipy.optimize import _lbfgsb
        major, minor = (int(v) for v in scipy.__version__.split(".")[:2])
        if (major, minor) < (1, 15):
            why = "scipy %s predates the routine's C translation (1.15)" % scipy.__version__
        elif not _probe_setulb(_lbfgsb.setulb):
            why = "scipy %s: the routine did not minimise the probe problem with the 1.15 argument list" % scipy.__version__
        else:
            found = _lbfgsb.setulb
    except Exception as exc:
        why = "%s: %s" % (type(exc).__name__, exc)
    _driver_probe.append(found)
    if found is None and warn:
        import warnings
        warnings.warn("gp_emulator_amd.perband: scipy.optimize._lbfgsb.setulb is not usable (%s); training falls back "
                      "to the threaded L-BFGS-B driver (same results, more launches)" % why, RuntimeWarning, stacklevel=2)
    return found


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


class _Population(object):
    """A set of L-BFGS-B instances advanced in lockstep: ``advance(results)`` hands the instances
    that asked last time their costs and gradients, steps every live instance through scipy's
    routine until it asks for a cost/gradient or stops, and returns the new requests and the
    instances that finished.  Lives in the calling process or in a worker process (``_worker``)."""

    factr, pgtol, m, maxls, maxfun, maxiter = 0.1, 1e-20, 10, 20, 15000, 15000

    def __init__(self, setulb, ids, starts):
        self.setulb = setulb
        self.n = int(np.asarray(starts).shape[1])
        self.live = [_Instance(e, k, x0) for (e, k), x0 in zip(ids, starts)]
        self.asking = []

    def advance(self, results):
        """results: (cost (A,), grad (A, n)) for the A requests returned by the previous call
        (None the first time).  Returns (thetas (A', n), e (A',), finished) where finished is a
        list of (e, k, cost, theta)."""
        finished = []
        if results is not None:
            c, g = results
            keep = []
            for j, it in enumerate(self.asking):
                it.nfev += 1
                if not np.isfinite(c[j]):
                    # a pivot <= 0: the reference's numpy path raises LinAlgError here and its
                    # optimiser loop gives the restart up with cost 9999 (:176-181)
                    finished.append((it.e, it.k, 9999.0, it.last_good))
                else:
                    it.f = float(c[j])
                    it.g[:] = g[j]
                    it.last_good[:] = it.x
                    keep.append(it)
            self.live = keep
        setulb, m, factr, pgtol, maxls = self.setulb, self.m, self.factr, self.pgtol, self.maxls
        asking = []
        for it in self.live:
            while True:
                setulb(m, it.x, it.low, it.up, it.nbd, it.f, it.g, factr, pgtol, it.wa, it.iwa,
                       it.task, it.lsave, it.isave, it.dsave, maxls, it.ln_task)
                task = it.task[0]
                if task == 3:                            # wants f and g at x
                    if it.nfev and it.x.tobytes() == it.last_good.tobytes():
                        continue                         # same point again: f, g are current
                    asking.append(it)                    # (what scipy's ScalarFunction cache does)
                    break
                if task == 1:                            # new iteration
                    it.nit += 1
                    if it.nit >= self.maxiter:
                        it.task[0], it.task[1] = 5, 504
                    elif it.nfev > self.maxfun:
                        it.task[0], it.task[1] = 5, 502
                    continue
                finished.append((it.e, it.k, it.f, it.x.copy()))   # converged / stopped
                break
        self.asking = self.live = asking
        th = np.stack([it.x for it in asking]) if asking else np.zeros((0, self.n))
        es = np.array([it.e for it in asking], dtype=np.int64)
        return th, es, finished


def _worker_main():
    """Body of a lockstep worker process (``python -c "from gp_emulator_amd.perband import
    _worker_main; _worker_main()"``): a _Population driven over its stdin / stdout with pickled
    messages.  It never touches the GPU -- the objective is evaluated by the parent, for all
    workers' requests at once.  A plain child process rather than ``multiprocessing``: a spawned
    multiprocessing child re-imports the caller's main script, which a library must not require
    to be import-safe."""
    import os
    import pickle
    import sys
    inp, out = sys.stdin.buffer, os.fdopen(os.dup(sys.stdout.fileno()), "wb")
    sys.stdout = sys.stderr                       # nothing but messages on the pipe
    try:
        ids, starts = pickle.load(inp)
        pop = _Population(_lockstep_driver(), ids, starts)
        results = None
        while True:
            pickle.dump(pop.advance(results), out, protocol=4)
            out.flush()
            results = pickle.load(inp)
            if results is None:
                return
    except EOFError:
        return
    except BaseException as exc:                  # report, the parent raises
        try:
            pickle.dump(RuntimeError("%s: %s" % (type(exc).__name__, exc)), out, protocol=4)
            out.flush()
        except Exception:
            pass


def _lockstep_processes(n_instances, processes):
    """How many worker processes step the optimisers.  scipy's routine holds the GIL and costs
    ~10 us of host time per evaluation -- 2e6 evaluations for 2101 bands x 5 restarts -- while the
    GPU needs 1.5 us for the same evaluation, so large jobs spread the stepping over processes;
    small ones are not worth the workers' start-up (an interpreter + numpy + scipy each)."""
    if processes is not None:
        return max(0, int(processes))
    if n_instances < 2048:
        return 0
    import os
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(0, min(8, avail - 1, n_instances // 512))


def _learn_lockstep(setulb, batch_fn, targets, starts, max_batch=4096, processes=0):
    """Every (emulator, restart) optimisation advances through scipy's own L-BFGS-B routine, no
    callbacks: each instance is stepped until it asks for a cost/gradient, then ALL the requests
    are evaluated in one batched call.  Iterates are those of ``fmin_l_bfgs_b(factr=0.1,
    pgtol=1e-20)`` (reference GaussianProcess.py:168-170) because it is the same routine fed the
    same numbers.  ``processes`` > 0: the instances are dealt to that many worker processes
    (child interpreters that import numpy and scipy and never touch the GPU); the parent only
    gathers their requests, runs the batched objective and scatters the answers."""
    E, n_tries, n = starts.shape
    ids = [(e, k) for e in range(E) for k in range(n_tries)]
    flat = starts.reshape(E * n_tries, n)
    cost = np.full((E, n_tries), np.inf)
    theta = np.zeros((E, n_tries, n))
    launches = evaluations = 0

    def evaluate(th, es):
        nonlocal launches, evaluations
        c = np.empty(len(th))
        g = np.empty((len(th), n))
        for s0 in range(0, len(th), max_batch):
            c[s0:s0 + max_batch], g[s0:s0 + max_batch] = batch_fn(th[s0:s0 + max_batch], targets[es[s0:s0 + max_batch]])
            launches += 1
        evaluations += len(th)
        return c, g

    def record(finished):
        for e, k, f, x in finished:
            cost[e, k], theta[e, k] = f, x

    if processes <= 0:
        pop = _Population(setulb, ids, flat)
        results = None
        while True:
            th, es, finished = pop.advance(results)
            record(finished)
            if not len(th):
                break
            results = evaluate(th, es)
        return cost, theta, launches, evaluations

    import os
    import pickle
    import subprocess
    import sys
    # the workers' BLAS must be single-threaded: the routine makes tiny LAPACK calls, and several
    # processes each spinning a thread team per call are hundreds of times slower than one thread
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "BLIS_NUM_THREADS"):
        env[k] = "1"
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    code = "from gp_emulator_amd.perband import _worker_main; _worker_main()"
    workers = []

    def send(proc, msg):
        pickle.dump(msg, proc.stdin, protocol=4)
        proc.stdin.flush()

    try:
        for w in range(processes):
            mine = list(range(w, len(ids), processes))          # dealt round-robin: even load
            proc = subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE,
                                    stdout=subprocess.PIPE, env=env)
            workers.append(proc)
            send(proc, ([ids[i] for i in mine], flat[mine]))
        # two groups of workers take turns: while the objective of one group's requests runs on
        # the GPU (the GIL is released in the call), the other group's optimisers are stepping
        groups = [list(range(0, processes, 2)), list(range(1, processes, 2))]
        while groups[0] or groups[1]:
            for gi in (0, 1):
                if not groups[gi]:
                    continue
                asked = []
                for w in groups[gi]:
                    try:
                        msg = pickle.load(workers[w].stdout)
                    except EOFError:
                        raise RuntimeError("lockstep worker %d exited (status %s)" % (w, workers[w].poll()))
                    if isinstance(msg, BaseException):
                        raise RuntimeError("lockstep worker %d failed: %s" % (w, msg))
                    th, es, finished = msg
                    record(finished)
                    asked.append((w, th, es))
                th_all = np.concatenate([th for _, th, _ in asked])
                es_all = np.concatenate([es for _, _, es in asked])
                c, g = evaluate(th_all, es_all) if len(th_all) else (np.zeros(0), np.zeros((0, n)))
                s0, still = 0, []
                for w, th, _ in asked:
                    k = len(th)
                    if k:
                        send(workers[w], (c[s0:s0 + k], g[s0:s0 + k]))
                        still.append(w)
                    else:
                        send(workers[w], None)              # nothing left in that worker
                    s0 += k
                groups[gi] = still
        for proc in workers:
            proc.wait(10)
    finally:
        for proc in workers:                            # the exact children started above
            for fh in (proc.stdin, proc.stdout):
                try:
                    fh.close()
                except Exception:
                    pass
            if proc.poll() is None:
                proc.terminate()
                try:
                    proc.wait(5)
                except Exception:
                    proc.kill()
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
                device=None, verbose=False, method="auto", processes=None):
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
        n_proc = _lockstep_processes(E * n_tries, processes)
        best_cost, best_theta, launches, evaluations = _learn_lockstep(setulb, batch_fn, tg_all, starts,
                                                                       processes=n_proc)
        pick = np.argmin(best_cost, axis=1)
        costs = best_cost[np.arange(E), pick]
        thetas = best_theta[np.arange(E), pick]
        _leave_set(gps, thetas, inputs, ctx)
        stats = {"launches": launches, "evaluations": evaluations, "threads": 1, "method": "lockstep",
                 "processes": n_proc}
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
