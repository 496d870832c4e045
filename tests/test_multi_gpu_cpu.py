"""The N>1 path on CPU: row sharding + host gather, and the world_size-2 gloo rank plumbing
(barrier, max-over-ranks timing, whole-job aggregation) that bench.py runs under
torch.distributed.run.  No GPU: the per-shard compute is a stand-in (the oracle)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synthetic_case
from oracle import gp_oracle

from gp_emulator_amd import GaussianProcess, multi_gpu


@pytest.mark.parametrize("M,G", [(0, 4), (1, 8), (7, 8), (8, 8), (1000, 3), (100000001, 8), (5, 1)])
def test_row_shards_partition(M, G):
    sh = multi_gpu.row_shards(M, G)
    assert len(sh) == G
    assert sh[0][0] == 0 and sh[-1][1] == M
    assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))           # contiguous, disjoint
    assert all(0 <= e - s <= -(-M // G) if M else e == s for s, e in sh)
    assert sum(e - s for s, e in sh) == M


def test_predict_sharded_gathers_disjoint_slices():
    g = synthetic_case("c1_n100_d5")
    gp = GaussianProcess(g["inputs"], [])
    gp.theta, gp.invQ, gp.invQt = g["theta"], g["invQ"], g["invQt"]
    seen = []

    def fake(device, rows):          # stand-in for the per-device HIP predict
        seen.append((device, rows.shape[0]))
        return gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"], g["invQt"], rows)

    mu, var, der = multi_gpu.predict_sharded(gp, g["testing"][:1003], devices=[0, 1, 2, 3],
                                             predict_fn=fake)
    assert sorted(seen) == [(0, 251), (1, 251), (2, 251), (3, 250)]
    assert gp_oracle.maxnorm_err(g["mu"][:1003], mu) < 1e-13
    assert gp_oracle.maxnorm_err(g["var"][:1003], var) < 1e-13
    assert gp_oracle.maxnorm_err(g["deriv"][:1003], der) < 1e-13

    def boom(device, rows):
        raise RuntimeError("device %d failed" % device)
    with pytest.raises(RuntimeError):
        multi_gpu.predict_sharded(gp, g["testing"][:10], devices=[0, 1], predict_fn=boom)


WORKER = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from gp_emulator_amd import multi_gpu
from oracle import gp_oracle
grp = multi_gpu.RankGroup()
inputs, testing, theta, invQ, invQt = gp_oracle.benchmark_inputs(1000 + grp.rank, 50, 4, 2000)
state = dict(n=0)
def step():
    gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing)
    state['n'] += 1
    if grp.rank == 1:
        time.sleep(0.02)          # the slow rank must set the reported time
dt = multi_gpu.timed_steps(grp, step, lambda: None, steps=5, warmup=2)
rows = grp.sum(5 * testing.shape[0])
assert state['n'] == 7
if grp.rank == 0:
    print(json.dumps(dict(world=grp.world, dt=dt, rows=rows)))
grp.close()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_world2_barrier_and_max_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2
    assert out["rows"] == 2 * 5 * 2000                 # whole-job units over all ranks
    assert out["dt"] >= 5 * 0.02                       # max over ranks: the slow rank's time


STRONG_WORKER = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
from gp_emulator_amd import multi_gpu
from oracle import gp_oracle
# the strong-scaling flow of bench.py --workload c4 --scaling strong, with the numpy path standing
# in for the GPU: a FIXED total of rows split over the ranks, every rank writing its shard
# straight into its slice of ONE shared host array (the host gather; no collective)
grp = multi_gpu.RankGroup()
total, D = 3001, 4
inputs, testing, theta, invQ, invQt = gp_oracle.benchmark_inputs(77, 50, D, total)   # same on every rank
lo, hi = multi_gpu.row_shards(total, grp.world)[grp.rank]
path = {path!r}
shared = multi_gpu.SharedOutputs(path, total, D, create=True) if grp.rank == 0 else None
grp.barrier()
if shared is None:
    shared = multi_gpu.SharedOutputs(path, total, D, create=False)
out = shared.views(lo, hi)
def step():
    out[0][:], out[1][:], out[2][:] = gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing[lo:hi])
dt = multi_gpu.timed_steps(grp, step, lambda: None, steps=2, warmup=1)
grp.barrier()
if grp.rank == 0:
    ref = gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing)
    # (BLAS sums a 1501-row block in a different order than a 3001-row one: equal to rounding)
    err = max(gp_oracle.maxnorm_err(a, b) for a, b in zip(ref, (shared.mu, shared.var, shared.deriv)))
    print(json.dumps(dict(world=grp.world, err=err, rows=int(hi - lo), dt=dt,
                          other_ranks_rows_present=bool(np.all(shared.mu[hi:] != 0)))))
grp.barrier()
del out
shared.close(unlink=grp.rank == 0)
grp.close()
"""


def test_gloo_world2_strong_scaling_host_gather(tmp_path):
    """Two gloo ranks on the CPU: disjoint row shards of a fixed total written into one shared
    host array equal the single-process result (rows are independent)."""
    path = str(tmp_path / "gathered.bin")
    script = tmp_path / "strong_worker.py"
    script.write_text(STRONG_WORKER.format(root=ROOT, path=path))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["world"] == 2 and out["rows"] == 1501
    assert out["err"] <= 1e-12 and out["other_ranks_rows_present"]
    assert not os.path.exists(path)


# ---------------------------------------------------------------------------------------------
# perband.learn_bands: the request-gathering runtime, driven by a numpy objective (no GPU)
# ---------------------------------------------------------------------------------------------
def _bands_problem(n_bands=5, n=30, d=2, seed=0):
    rs = np.random.RandomState(seed)
    X = rs.random_sample((n, d))
    bands = [np.sin(X.sum(1) * (1 + 0.3 * e)) + 0.05 * rs.standard_normal(n) for e in range(n_bands)]
    return X, bands


def _numpy_objective(X, calls):
    from oracle import gp_oracle

    def fn(thetas, targets):
        calls.append(len(thetas))
        c = np.array([gp_oracle.loglikelihood(X, tg, th) for th, tg in zip(thetas, targets)])
        g = np.stack([gp_oracle.partial_devs(X, tg, th) for th, tg in zip(thetas, targets)])
        return c, g
    return fn


@pytest.mark.parametrize("method", ["lockstep", "threads"])
def test_learn_bands_matches_a_loop_of_learn_hyperparameters(method):
    """Same starting points (drawn in the same order from numpy.random), same optimiser, batched
    objective: every band ends at the optimum the reference-style loop finds, and the requests
    really were gathered (several per call of the objective).  "lockstep" drives scipy's
    reverse-communication routine directly, "threads" gathers the requests of fmin_l_bfgs_b
    calls running in threads."""
    import warnings
    from gp_emulator_amd import GaussianProcess, perband
    if method == "lockstep" and perband._lockstep_driver() is None:
        pytest.skip("this scipy does not expose the reverse-communication routine")
    X, bands = _bands_problem()
    gps = [GaussianProcess(X, t) for t in bands]
    calls = []
    np.random.seed(11)
    costs, thetas, stats = perband.learn_bands(gps, n_tries=2, concurrency=4, method=method,
                                               batch_fn=_numpy_objective(X, calls))
    assert stats["method"] == method
    assert stats["evaluations"] == sum(calls) and stats["launches"] == len(calls)
    if method == "threads":
        assert stats["threads"] == 4 and max(calls) == 4 and sum(calls) / len(calls) > 2.0
    else:
        assert max(calls) == 10 and sum(calls) / len(calls) > 4.0      # 5 bands x 2 starts at once
    np.random.seed(11)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for e, t in enumerate(bands):
            ref = GaussianProcess(X, t)
            c, th = ref.learn_hyperparameters(n_tries=2)
            assert abs(c - costs[e]) <= 1e-6 * max(1.0, abs(c))
            # the emulator is left set to its optimum, ready to predict
            assert np.allclose(gps[e].theta, thetas[e]) and gps[e].invQ.shape == (30, 30)
            assert np.allclose(gps[e].predict(X[:5])[0], ref.predict(X[:5])[0], atol=1e-5)


def test_learn_bands_edge_cases():
    from gp_emulator_amd import GaussianProcess, perband
    X, bands = _bands_problem(n_bands=3)
    gps = [GaussianProcess(X, t) for t in bands]
    with pytest.raises(ValueError):
        perband.learn_bands([], batch_fn=lambda a, b: None)
    with pytest.raises(ValueError):
        perband.learn_bands(gps, is_gpu=False)                       # no silent CPU path
    with pytest.raises(ValueError):
        perband.learn_bands(gps + [GaussianProcess(X + 1.0, bands[0])], batch_fn=lambda a, b: None)

    def broken(thetas, targets):
        raise RuntimeError("objective failed")
    for method in ("threads", "auto"):
        with pytest.raises(RuntimeError):                             # every thread is released
            perband.learn_bands(gps, n_tries=2, concurrency=3, batch_fn=broken, method=method)
    # more threads than problems, explicit starts, one try
    calls = []
    starts = np.zeros((3, 1, 4))
    costs, thetas, stats = perband.learn_bands(gps, concurrency=64, starts=starts, method="threads",
                                               batch_fn=_numpy_objective(X, calls))
    assert stats["threads"] == 3 and costs.shape == (3,) and thetas.shape == (3, 4)
    with pytest.raises(ValueError):
        perband.learn_bands(gps, starts=starts, method="bogus", batch_fn=_numpy_objective(X, calls))


def test_private_scipy_routine_is_probed_and_falls_back_with_a_warning(monkeypatch):
    """scipy.optimize._lbfgsb.setulb is private: a scipy whose routine takes other arguments (or answers with other
    task codes) is found out by the probe run, learn_bands(method="auto") warns once and trains with the threaded
    driver to the same optimum, method="lockstep" refuses."""
    from gp_emulator_amd import GaussianProcess, perband
    from scipy.optimize import _lbfgsb
    real = perband._lockstep_driver(warn=False)
    if real is not None:
        assert perband._probe_setulb(real)
    assert not perband._probe_setulb(lambda *a: None)                 # a routine that does nothing

    def other_signature(m, x, low, up, nbd, f, g, factr, pgtol, wa, iwa, task, iprint, lsave, isave, dsave, maxls):
        raise AssertionError("not called with 17 arguments")
    del perband._driver_probe[:]
    try:
        with monkeypatch.context() as mp:                             # (scipy's own fmin_l_bfgs_b calls it too: patched
            mp.setattr(_lbfgsb, "setulb", other_signature)            # for the probe only)
            with pytest.warns(RuntimeWarning, match="falls back to the threaded"):
                assert perband._lockstep_driver() is None
        X, bands = _bands_problem(n_bands=2)
        gps = [GaussianProcess(X, t) for t in bands]
        np.random.seed(3)
        costs, thetas, stats = perband.learn_bands(gps, n_tries=1, concurrency=2, method="auto",
                                                   batch_fn=_numpy_objective(X, []))
        assert stats["method"] == "threads" and np.all(np.isfinite(costs))
        with pytest.raises(RuntimeError):
            perband.learn_bands(gps, n_tries=1, method="lockstep", batch_fn=_numpy_objective(X, []))
    finally:
        del perband._driver_probe[:]                                  # the next test probes the real routine again


def test_learn_bands_lockstep_equals_threads_bit_for_bit():
    """Both drivers feed the same numbers to the same L-BFGS-B routine: identical thetas, and the
    same number of objective evaluations (repeated requests at an unchanged point are served
    from the last result, as scipy's own wrapper does)."""
    from gp_emulator_amd import GaussianProcess, perband
    if perband._lockstep_driver() is None:
        pytest.skip("this scipy does not expose the reverse-communication routine")
    X, bands = _bands_problem(n_bands=4)
    out = {}
    for method in ("lockstep", "threads"):
        gps = [GaussianProcess(X, t) for t in bands]
        np.random.seed(5)
        out[method] = perband.learn_bands(gps, n_tries=2, concurrency=3, method=method,
                                          batch_fn=_numpy_objective(X, []))
    assert np.array_equal(out["lockstep"][0], out["threads"][0])
    assert np.array_equal(out["lockstep"][1], out["threads"][1])
    assert out["lockstep"][2]["evaluations"] == out["threads"][2]["evaluations"]


def test_learn_bands_worker_processes_change_nothing_but_the_wall_clock():
    """processes > 0 deals the optimisers to child interpreters (scipy's routine holds the GIL);
    the parent gathers their requests into the same batched objective.  Same iterates, same
    evaluation count, same optimum -- and a failing objective still surfaces in the caller."""
    from gp_emulator_amd import GaussianProcess, perband
    if perband._lockstep_driver() is None:
        pytest.skip("this scipy does not expose the reverse-communication routine")
    assert perband._lockstep_processes(100, None) == 0 and perband._lockstep_processes(100, 3) == 3
    X, bands = _bands_problem(n_bands=4)
    out = {}
    for procs in (0, 3):
        gps = [GaussianProcess(X, t) for t in bands]
        np.random.seed(5)
        calls = []
        out[procs] = perband.learn_bands(gps, n_tries=2, method="lockstep", processes=procs,
                                         batch_fn=_numpy_objective(X, calls))
        # one batch per round, or one per group of workers (two groups take turns): 8 = 5 + 3
        assert out[procs][2]["processes"] == procs and max(calls) == (8 if procs == 0 else 5)
    assert np.array_equal(out[0][0], out[3][0]) and np.array_equal(out[0][1], out[3][1])
    assert out[0][2]["evaluations"] == out[3][2]["evaluations"]

    def broken(thetas, targets):
        raise RuntimeError("objective failed")
    with pytest.raises(RuntimeError):
        perband.learn_bands([GaussianProcess(X, t) for t in bands], n_tries=2, method="lockstep",
                            processes=2, batch_fn=broken)


def test_predict_bands_shards_emulators_over_devices():
    """BASELINE config 3's other partition: the EMULATORS are cut into contiguous blocks, one per
    device, the shared test rows go to every device, and each block lands in its own slice of the
    outputs (host gather, no collective).  The HIP path is replaced by a numpy one here."""
    from gp_emulator_amd import GaussianProcess, perband
    X, bands = _bands_problem(n_bands=7)
    gps = []
    for k, t in enumerate(bands):
        gp = GaussianProcess(X, t)
        gp._set_params(np.array([0.3 + 0.1 * k] * X.shape[1] + [0.5, -4.0]))
        gps.append(gp)
    testing = np.random.RandomState(2).random_sample((33, X.shape[1]))
    seen = []

    def fake(dev, part, rows):
        seen.append((dev, len(part)))
        out = [gp.predict(rows) for gp in part]
        return np.stack([o[0] for o in out]), np.stack([o[1] for o in out]), np.stack([o[2] for o in out])
    mu, var, der = perband.predict_bands(gps, testing, devices=[0, 1, 2], predict_fn=fake)
    assert sorted(seen) == [(0, 3), (1, 3), (2, 1)]
    for e, gp in enumerate(gps):
        m, v, d = gp.predict(testing)
        assert np.array_equal(mu[e], m) and np.array_equal(var[e], v) and np.array_equal(der[e], d)
    mu2 = perband.predict_bands(gps[:2], testing, devices=[0, 1, 2, 3], predict_fn=fake)[0]   # more devices than bands
    assert np.array_equal(mu2, mu[:2])

    def broken(dev, part, rows):
        raise RuntimeError("device %d failed" % dev)
    with pytest.raises(RuntimeError):
        perband.predict_bands(gps, testing, devices=[0, 1], predict_fn=broken)
    with pytest.raises(ValueError):
        perband.predict_bands(gps + [GaussianProcess(X + 1.0, bands[0])], testing, devices=[0], predict_fn=fake)


def test_output_pool_reuses_only_unreferenced_buffers():
    """_lib.OutputPool hands a buffer out again only when nothing but the pool refers to it: results
    a caller still holds (or views of them) are never overwritten by a later predict."""
    from gp_emulator_amd._lib import OutputPool
    pool = OutputPool(max_item=64 << 20, max_total=40 << 20)
    a = pool.take((2 << 20,), np.float64)                 # 16 MB
    addr = a.ctypes.data
    b = pool.take((2 << 20,), np.float64)
    assert b.ctypes.data != addr                          # a is alive
    del a
    c = pool.take((1 << 20, 2), np.float64)               # same bytes, other shape
    assert c.ctypes.data == addr and c.shape == (1 << 20, 2)
    view = c[:10]
    del c
    d = pool.take((2 << 20,), np.float64)
    assert d.ctypes.data != addr                          # a view of c is alive
    view[:] = 7.0
    assert np.all(view == 7.0)
    assert sum(x.nbytes for x in pool.items) <= 40 << 20  # over budget: buffers are forgotten, not reused
    small = pool.take((100,), np.float64)
    assert small.base is None                             # small and oversize requests bypass the pool
    big = pool.take((9 << 20,), np.float64)               # 72 MB > max_item
    assert big.base is None


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N without a launcher: it starts its own ranks, and never reports a GPU count it did not run on
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("extra", [[], ["--workload", "c4", "--scaling", "strong"]])
def test_bench_gpus_2_starts_its_own_ranks(extra):
    """`python bench.py --gpus 2` with no torch.distributed.run around it: two child ranks rendezvous over gloo,
    rank 0's line says n_gpus == 2 (GP_BENCH_DRY_RUN=1: rank plumbing only, no device work, labelled so)."""
    env = dict(os.environ, GP_BENCH_DRY_RUN="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3
    assert "dry run" in out["data"] and out["value"] == 0.0      # never mistaken for a measurement


def test_bench_gpus_mismatch_and_no_gpu_exit_nonzero():
    """A launcher's WORLD_SIZE that contradicts --gpus is an error, and without a GPU the self-launched job fails as
    a whole (no line at all) instead of printing a line for fewer GPUs."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    from gp_emulator_amd import _lib
    try:
        n = _lib.device_count()
    except _lib.GpuPredictUnavailable:
        n = 0
    if n == 0:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
        assert r.returncode != 0
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
