#!/usr/bin/env python3
"""bench.py -- test-points/sec for predict(mean+var+grad), N_train=250, D=11 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the fused predict kernel over one batch of M = 1e6 synthetic test
rows (BASELINE config 2, "PROSAIL single-band": N_train=250, D=11, fp64, mean+var+grad)
that is ALREADY RESIDENT in HBM; outputs stay in HBM.  With N ranks every rank owns its own
1e6-row block and its own GPU: independent row blocks, no collective on the data path
(SURVEY.md section 8e), so scaling is weak.  torch is imported only for the multi-rank
barrier / max-reduce (gloo); the product path is ctypes -> libgp_predict_hip.so.

Extra objects on the JSON line (flat scalars only inside each: the driver's record keeps those):
  roofline      the fused kernel against the FP64 pipe peak (matrix = vector on MI355X; the
                binding roofline, arithmetic intensity ~746 flop/B).  ``achieved`` / ``frac`` count
                the flops the kernel EXECUTES (matrix instructions issued + phase-A arithmetic),
                so frac <= 1; ``algorithmic_ratio`` is the un-halved count of SURVEY.md 8d over
                the same peak (it exceeds 1: symmetric folding halves the variance work);
                ``hbm_gbps`` / ``hbm_frac`` put the same launch against the 8 TB/s HBM roofline
                (the figure BASELINE.json's north star asks for); ``e2e_points_per_s`` is
                gp.predict(is_gpu=True) from host numpy arrays to host numpy arrays, what the
                reference's tests/benchmark.py:41-44 times (never ``value``).
  cpu_baseline  the numpy path (oracle = py3 restatement of the reference's cpu_predict) on this
                box's host cores, rank 0, N=1 only: ``value`` as the reference runs it (one
                process, BLAS threads = the cpu share), ``all_cores_value`` one single-threaded
                worker process per cpu of the share over disjoint row shards.

``--workload c4 --scaling strong`` is the BASELINE configs[3] run proper: a FIXED total of
--total-rows (default 1e8) test rows split over the ranks, every rank predicting its shard from
host arrays into its slice of ONE shared host array (/dev/shm), i.e. the host gather is timed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_TRAIN, N_INPUTS, N_TEST = 250, 11, 1000000
# algorithmic work per test point, fp64, N=250, D=11
FLOP_PER_POINT = 143262          # 3ND + 3N + 2N + (2N^2 + 2N + 1) + (3ND + D)
BYTES_PER_POINT = 192            # read D*8 + write (2 + D)*8
# FP64 matrix = FP64 vector peak of MI355X (AMD datasheet figure quoted in SURVEY.md 8d;
# MI355X_MICROARCH.md lists no fp64 row): 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz
PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def synthetic_inputs(seed, n_train, n_inputs, n_predict):
    """The seeded form of the reference benchmark's inputs (tests/benchmark.py:11-15,28-29):
    U[0,1) training inputs, test rows, theta, invQ (deliberately non-symmetric), invQt, drawn
    in that order.  Same recipe as the fixtures' (oracle.gp_oracle.benchmark_inputs), restated
    here so that the timed path depends on nothing under oracle/."""
    rs = np.random.RandomState(seed)
    inputs = rs.random_sample((n_train, n_inputs))
    testing = rs.random_sample((n_predict, n_inputs))
    theta = rs.random_sample(n_inputs + 2)
    invQ = rs.random_sample((n_train, n_train))
    invQt = rs.random_sample(n_train)
    return inputs, testing, theta, invQ, invQt


def maxnorm_err(ref, got):
    """max|ref-got| / max|ref| (tests/benchmark.py:51-53)."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    return float(np.max(np.abs(ref - got)) / np.max(np.abs(ref)))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5", "mv", "train"],
                    help="c2 = the metric's config (default); c3/c4/c5 = the other BASELINE configs")
    ap.add_argument("--n-test", type=int, default=0, help="rows per GPU per step (0 = workload default)")
    ap.add_argument("--emulators", type=int, default=2101, help="emulators in the c3 batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="timing-only ablation builds (tools/ab_bench.py): do not stop on a parity failure")
    ap.add_argument("--cpu-sample", type=int, default=1000000)
    ap.add_argument("--cpu-procs", type=int, default=0,
                    help="worker processes of the all-cores CPU leg (0 = every cpu this process may use)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-host leg")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="strong (c4 only): fixed --total-rows split over the ranks, host arrays in, "
                         "shared host array out")
    ap.add_argument("--total-rows", type=int, default=100000000)
    ap.add_argument("--cpu-worker", default="", help=argparse.SUPPRESS)
    return ap.parse_args()


def cpu_quota():
    """cpus this process may really use: the affinity mask, cut down by the cgroup's CFS quota when there is
    one (cpu.max = "<quota> <period>").  On the 1-GPU MI355X box of this pool: 256 cpus in the mask, a quota of
    16 -- more worker processes than that only share those 16."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return avail, quota


def cpu_share():
    """Worker count of the CPU legs: every cpu this process may use (mask and cgroup quota), at most 256."""
    avail, quota = cpu_quota()
    n = avail if quota is None else min(avail, max(1, int(round(quota))))
    return max(1, min(n, 256))


def host_description():
    """Host facts BASELINE.md section 3 asks for beside a CPU number."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    avail, quota = cpu_quota()
    out = {"host_cpu_model": model, "host_cpus": int(os.cpu_count() or 0), "cpus_in_affinity_mask": int(avail),
           "cgroup_cpu_quota": quota, "numpy": np.__version__}
    try:
        import scipy
        out["scipy"] = scipy.__version__
    except ImportError:
        pass
    try:
        from threadpoolctl import threadpool_info
        out["blas"] = "; ".join("%s %s (%s threads)" % (i.get("internal_api"), i.get("version"), i.get("num_threads"))
                                for i in threadpool_info() if i.get("user_api") == "blas")
    except ImportError:
        pass
    return out


def cpu_worker(spec):
    """One worker of the all-cores leg (its own process, BLAS pinned to one thread by the
    parent's environment): ``seed,rows`` -> predicts its own seeded shard, prints the seconds."""
    from oracle import gp_oracle
    seed, rows = (int(x) for x in spec.split(","))
    inputs, _, theta, invQ, invQt = synthetic_inputs(12345, N_TRAIN, N_INPUTS, 1)
    testing = np.random.RandomState(seed).random_sample((rows, N_INPUTS))
    gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing[:1000])      # warm up
    t0 = time.perf_counter()
    gp_oracle.cpu_predict_blocked(inputs, theta, invQ, invQt, testing, block=20000)
    print("%.6f" % (time.perf_counter() - t0), flush=True)


def cpu_baseline(sample_rows, procs):
    """numpy path on the host (both variants).  Runs BEFORE anything touches
    the GPU, so starting worker processes is an ordinary fork + exec.
    (1) as the reference runs it: the oracle's restatement of GaussianProcess.cpu_predict in ONE
        process (scipy cdist + numpy; BLAS threads = the cpu share for the np.dot calls,
        everything else single-threaded);
    (2) all cores: one worker process per cpu of the share, BLAS single-threaded, disjoint
        seeded row shards; rows of all workers / wall time from first start to last exit."""
    import subprocess
    from oracle import gp_oracle
    from threadpoolctl import threadpool_limits
    inputs, testing, theta, invQ, invQt = synthetic_inputs(12345, N_TRAIN, N_INPUTS, sample_rows)
    blas_threads = cpu_share()
    with threadpool_limits(limits=blas_threads):
        gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing[:2000])   # warm up
        t0 = time.perf_counter()
        gp_oracle.cpu_predict_blocked(inputs, theta, invQ, invQt, testing, block=50000)
        dt = time.perf_counter() - t0
    del testing
    procs = procs or blas_threads
    rows_each = max(1000, sample_rows // procs)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t0 = time.perf_counter()
    ws = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                            "%d,%d" % (777 + w, rows_each)], env=env, stdout=subprocess.PIPE, text=True)
          for w in range(procs)]
    inner = [float(w.communicate()[0].strip().splitlines()[-1]) for w in ws]
    wall = time.perf_counter() - t0
    if any(w.returncode != 0 for w in ws):
        raise SystemExit("cpu_baseline worker failed")
    return dict(host_description(), **{"value": sample_rows / dt, "unit": "test-points/s",
            "cores": int(blas_threads), "kind": "port",
            "sample": "%d rows of the same N=250, D=11 workload in 50k-row blocks, %.1f s; "
                      "numpy+scipy path (oracle/gp_oracle.py), BLAS threads=%s (every cpu this process may "
                      "use; the host has %d), elementwise/cdist/exp single-threaded as in the reference"
                      % (sample_rows, dt, blas_threads, os.cpu_count() or 0),
            "all_cores_value": procs * rows_each / max(inner),
            "all_cores_procs": int(procs),
            "all_cores_sample": "%d single-threaded worker processes (= every cpu this process may use: affinity "
                                "mask cut by the cgroup quota) x %d rows in 20k-row blocks, slowest "
                                "worker %.1f s (%.1f s wall with interpreter start-up)"
                                % (procs, rows_each, max(inner), wall)})


WORKLOADS = {
    # name: (n_train, n_inputs, default rows per GPU per step, kind)
    "c2": (250, 11, 1000000, "predict"),     # BASELINE configs[1] -- the metric's config
    "c3": (250, 11, 100000, "batch"),        # configs[2]: 2101 emulators, shared test rows
    "c4": (300, 11, 12500000, "predict"),    # configs[3]: 1e8 rows over 8 GPUs = 1.25e7 each
    "c5": (300, 16, 1000000, "hessian"),     # configs[4]: full DxD Hessian
}


def flop_per_point(N, D, kind):
    """3ND [dist] + 3N + 2N [mean] + (2N^2+2N+1) [var] + (3ND+D) [grad];
    Hessian as the reference writes it: 3ND + 3N + 7ND^2."""
    if kind == "hessian":
        return 3 * N * D + 3 * N + 7 * N * D * D
    return 3 * N * D + 3 * N + 2 * N + (2 * N * N + 2 * N + 1) + (3 * N * D + D)


def bench_reconstruct(a):
    """--workload mv: the MultivariateEmulator reconstruction for
    M = 1e5 test rows, 12 PCs, D = 10, 2101 bands: fwd (M, 2101) and Jacobian (M, 10, 2101)
    from per-PC outputs resident in HBM.  HBM-write-bound: 8 B written per 12 fma."""
    from gp_emulator_amd import _lib
    ctx = _lib.Context(0)
    info = ctx.device_info()
    M, P, D, B = a.n_test or 100000, 12, 10, 2101
    dt = np.float64 if a.precision == "f64" else np.float32
    isz = np.dtype(dt).itemsize
    rs = np.random.RandomState(3)
    basis = rs.standard_normal((P, B)).astype(dt)
    mu = rs.standard_normal((P, M)).astype(dt)
    grad = rs.standard_normal((P, M * D)).astype(dt)
    d_b, d_mu, d_g = ctx.to_device(basis), ctx.to_device(mu), ctx.to_device(grad)
    d_f, d_j = ctx.malloc(M * B * isz), ctx.malloc(M * D * B * isz)

    def step():
        ctx.reconstruct_device(dt, d_b, d_mu, d_f, M, P, B)
        ctx.reconstruct_device(dt, d_b, d_g, d_j, M * D, P, B)
    for _ in range(a.warmup):
        step()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    ctx.record(e0)
    for _ in range(a.steps):
        step()
    ctx.record(e1)
    ctx.synchronize()
    dt_wall = time.perf_counter() - t0
    kern_s = ctx.elapsed_ms(e0, e1) * 1e-3 / a.steps
    idx = rs.choice(M, 64, replace=False)
    got = ctx.to_host(d_f, (M, B), dt)[idx]
    ref = mu.astype(np.float64)[:, idx].T @ basis.astype(np.float64)
    err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    tol = 1e-12 if a.precision == "f64" else 1e-5
    if not err <= tol:
        raise SystemExit("reconstruct parity failed: %g" % err)
    bytes_step = (M * B + M * D * B) * isz            # written; coefficients are 0.06 % of it
    gbps = bytes_step / kern_s / 1e9
    out = {"metric": "test rows/sec for MultivariateEmulator reconstruction (fwd + Jacobian), 12 PCs, D=10, 2101 bands",
           "value": a.steps * M / dt_wall, "unit": "test-rows/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": dt_wall / a.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
           "config": {"workload": "MultivariateEmulator reconstruction (gp_emulator/multivariate_gp.py:214-218 "
                                  "for M rows): n_test=%d, n_pcs=12, n_inputs=10, n_bands=2101; per-PC means "
                                  "and gradients resident in HBM" % M,
                      "device": info["name"]},
           "roofline": {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": gbps / PEAK_HBM_GBPS, "traffic": None,
                        "kernel": "reconstruct_kernel<%s,12> x2" % ("double" if a.precision == "f64" else "float"),
                        "kernel_ms": kern_s * 1e3, "bytes_per_row": (B + D * B) * isz},
           "parity": {"e_fwd": err, "tol": tol, "checked_rows": 64}}
    print(json.dumps(out), flush=True)
    for p_ in (d_b, d_mu, d_g, d_f, d_j):
        ctx.free(p_)
    return out


def bench_train(a):
    """--workload train: the training objective for E = 2101 per-band
    emulators at once: cost + gradient + invQ + invQt of N_train = 250, D = 10 each, one
    launch per step; the numpy path (oracle) is timed on a few sets beside it."""
    from gp_emulator_amd import _lib
    from oracle import gp_oracle
    ctx = _lib.Context(0)
    info = ctx.device_info()
    E, N, D = a.emulators, 250, 10
    rs = np.random.RandomState(11)
    inputs = rs.random_sample((N, D))
    targets = np.sin(3 * inputs[:, 0])[None, :] * (1 + 0.1 * rs.standard_normal((E, 1))) \
        + 0.05 * rs.standard_normal((E, N))
    thetas = 0.5 * rs.standard_normal((E, D + 2))
    thetas[:, D + 1] -= 4.0
    for _ in range(a.warmup):
        ctx.likelihood_batch(thetas, inputs, targets)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cost, grad = ctx.likelihood_batch(thetas, inputs, targets)
    dt = (time.perf_counter() - t0) / a.steps
    mfma_flop = ((N + 7) // 8) * 136 * 2 * 2048     # matrix-core flops of one evaluation (N <= 256)
    n_cpu = min(8, E)
    t1 = time.perf_counter()
    for e in range(n_cpu):
        c = gp_oracle.loglikelihood(inputs, targets[e], thetas[e])
        g = gp_oracle.partial_devs(inputs, targets[e], thetas[e])
        assert abs(c - cost[e]) <= 1e-8 * abs(c)
        assert np.max(np.abs(g - grad[e])) <= 1e-5 * np.max(np.abs(g))
    cpu_per = (time.perf_counter() - t1) / n_cpu
    out = {"metric": "hyper-parameter sets/sec for loglikelihood + partial_devs, N_train=250 D=10",
           "value": E / dt, "unit": "theta-evaluations/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "training objective for %d per-band emulators (N_train=250, D=10): cost, "
                                  "gradient, invQ, invQt per theta; host arrays in and out" % E,
                      "device": info["name"]},
           "roofline": {"bound": "mfma", "achieved": E * mfma_flop / dt / 1e12, "peak": PEAK_FP64_TFLOPS,
                        "unit": "TFLOP/s", "frac": E * mfma_flop / dt / 1e12 / PEAK_FP64_TFLOPS,
                        "traffic": None, "kernel": "likelihood_mfma_kernel<12>",
                        "executed_flop_per_theta": mfma_flop,
                        "note": "register-resident Gauss-Jordan (gp_train_mfma_kernel.hpp): ceil(N/8) passes x 136 "
                                "lower-triangle tiles x 2 fp64 matrix instructions of 2048 flop; achieved = those flops / "
                                "wall time of the whole evaluation (Q build, panel steps, invQt, gradient and the host "
                                "copies of theta / cost / gradient included) -- the evaluation is bound by the serial "
                                "panel steps and barriers of each pass, not by the matrix pipe"},
           "cpu_baseline": {"value": 1.0 / cpu_per, "unit": "theta-evaluations/s", "cores": int(os.cpu_count() or 1),
                            "kind": "port", "sample": "%d evaluations of the numpy path (oracle)" % n_cpu},
           "parity": {"checked_sets": n_cpu, "tol_cost": 1e-8, "tol_grad": 1e-5}}
    print(json.dumps(out), flush=True)
    return out


def executed_flop_per_point(N, D, kind, minfo, hess_mfma):
    """Flops the kernel EXECUTES per test point on the fp64 (or fp32) pipe.
    predict: matrix instructions issued -- k-steps (4 I + s) < NK of block pairs I >= J, 2048 flop
    each per 16 rows -- plus phase A's arithmetic (everything of SURVEY.md 8d but the variance).
    Hessian (matrix core): 4 x 4 blocks bi <= bj of the D x D matrix, 4 NB instructions per block and
    16-row tile, plus phase A (kernel row, weights, s, G_d); Hessian (VALU): the upper triangle."""
    nb, kd = minfo["kernel_nb"], minfo["kernel_d"]
    if kind == "hessian":
        phase_a = 3 * N * D + 3 * N + 2 * N + 2 * N * D
        if hess_mfma:
            nb4 = (kd + 3) // 4
            if hess_mfma == "win":      # s and G_d ride on the matrix core's spare slots: not in phase A any more
                phase_a -= 2 * N + 2 * N * D
            ksteps = 4 * nb
            if hess_mfma == "win" and nb in (16, 19) and (N + 3) // 4 == 4 * nb - 1:
                ksteps -= 1             # the instance that does not issue the all-padding last k-step (N = 300: 75 of 76)
            return nb4 * (nb4 + 1) // 2 * ksteps * 2048 // 16 + phase_a
        return 2 * 16 * ((N + 15) // 16) * kd * (kd + 1) // 2 + phase_a
    nk = minfo["kernel_nk"]
    issued = sum(1 for J in range(nb) for I in range(J, nb) for s_ in range(4) if 4 * I + s_ < nk)
    phase_a = 3 * N * D + 3 * N + 2 * N + (3 * N * D + D)
    return issued * 2048 // 16 + phase_a


def sample_parity(ctx, a, kind, bufs_out, inputs, testing, params, M, D, E, dtype):
    """Spot check of what was timed against the oracle: sampled rows of the first, a middle and
    the last emulator, mean, variance AND gradient (rows are copied back one by one, so config 3's
    18 GB of gradients never cross PCIe whole)."""
    from oracle import gp_oracle
    from threadpoolctl import threadpool_limits
    rs = np.random.RandomState(5)
    isz = np.dtype(dtype).itemsize
    # (one BLAS thread: a pool of spinning BLAS workers would eat into the cpu share the
    # host-to-host leg's helper threads run on)
    limit = threadpool_limits(limits=1)
    if kind == "hessian":
        idx = np.sort(rs.choice(M, 256, replace=False))
        got = np.stack([ctx.to_host_at(bufs_out[0], i * D * D * isz, (D, D), dtype) for i in idx])
        theta, invQ, invQt = params(0)
        ref = gp_oracle.hessian(inputs, theta, invQt, testing[idx])
        return {"e_hess": maxnorm_err(ref, got)}, len(idx)
    d_mu, d_var, d_der = bufs_out
    n_rows = 2048 if E == 1 else 256
    idx = np.sort(rs.choice(M, n_rows, replace=False))
    worst = {"e_mu": 0.0, "e_var": 0.0, "e_deriv": 0.0}
    for e in sorted(set([0, E // 2, E - 1])):
        theta, invQ, invQt = params(e)
        mu = np.array([ctx.to_host_at(d_mu, (e * M + i) * isz, (), dtype) for i in idx])
        var = np.array([ctx.to_host_at(d_var, (e * M + i) * isz, (), dtype) for i in idx])
        der = np.stack([ctx.to_host_at(d_der, (e * M + i) * D * isz, (D,), dtype) for i in idx])
        ref = gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing[idx])
        for k, r, g in zip(("e_mu", "e_var", "e_deriv"), ref, (mu, var, der)):
            worst[k] = max(worst[k], maxnorm_err(r, g))
    return worst, n_rows * len(set([0, E // 2, E - 1]))


def e2e_leg(N, D, M, inputs, testing, theta, invQ, invQt, dtype, grp=None):
    """gp.predict(is_gpu=True) from host numpy arrays to host numpy arrays (default threshold),
    the call tests/benchmark.py:41-44 times; median of 10 calls after 3 warm-up calls.  With more
    than one rank every rank makes the same call on its own rows at the same moment (a barrier in
    front of each call) and a call takes as long as its slowest rank: the aggregate is what the
    host's DRAM and PCIe roots sustain with all the GPUs streaming at once."""
    from gp_emulator_amd import GaussianProcess
    gp = GaussianProcess(inputs, [])
    gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
    world = 1 if grp is None else grp.world
    ts = []
    for k in range(13):
        if world > 1:
            grp.barrier()
        t0 = time.perf_counter()
        out = gp.predict(testing, is_gpu=True, precision=dtype)
        dt = time.perf_counter() - t0
        ts.append(grp.max(dt) if world > 1 else dt)
        del out
    first, steady = ts[0], float(np.median(ts[3:]))
    # the same call for a caller that keeps its rows and result arrays page-locked (gp_emulator_amd.pinned_empty):
    # the library then copies straight between them and the device, no staging, no host copies
    pinned = None
    if np.dtype(dtype) == np.float64:
        import gp_emulator_amd
        t_pin = gp_emulator_amd.pinned_empty((M, D), dtype)
        t_pin[...] = testing
        pout = (gp_emulator_amd.pinned_empty((M,), dtype), gp_emulator_amd.pinned_empty((M,), dtype),
                gp_emulator_amd.pinned_empty((M, D), dtype))
        tp = []
        for k in range(8):
            if world > 1:
                grp.barrier()
            t0 = time.perf_counter()
            gp.predict(t_pin, is_gpu=True, precision=dtype, out=pout)
            dt = time.perf_counter() - t0
            tp.append(grp.max(dt) if world > 1 else dt)
        pinned = float(np.median(tp[2:]))
        del t_pin, pout
    return world * M / steady, steady, first, pinned


def bench_strong(a, grp):
    """--workload c4 --scaling strong: BASELINE configs[3] as the north star states it -- a fixed
    total of test rows row-sharded over the ranks, constants replicated, every rank predicting its
    shard host-to-host through the slab pipeline into ITS slice of one shared host array
    (/dev/shm): the gather is the D2H staging copy itself, there is no collective."""
    from gp_emulator_amd import _lib, multi_gpu
    rank, world = grp.rank, grp.world
    N, D = 300, 11
    total = int(a.total_rows)
    lo, hi = multi_gpu.row_shards(total, world)[rank]
    M = hi - lo
    dtype = np.float64 if a.precision == "f64" else np.float32
    ndev, n_gpus_used = devices_for(world)
    near = _lib.bind_near_device(grp.local_rank % ndev)      # before the shard's rows are generated
    inputs, _, theta, invQ, invQt = synthetic_inputs(1000, N, D, 1)
    testing = np.empty((M, D))
    for s0 in range(0, M, 1 << 20):                    # per-shard seeded generation, bounded temporaries
        n = min(1 << 20, M - s0)
        testing[s0:s0 + n] = np.random.RandomState(2000 + (lo + s0) // (1 << 20)).random_sample((n, D))
    path = os.environ.get("GP_BENCH_SHM", "/dev/shm/gp_bench_c4_%s.bin" % os.environ.get("MASTER_PORT", "0"))
    if rank == 0:
        shared = multi_gpu.SharedOutputs(path, total, D, create=True)
    grp.barrier()
    if rank != 0:
        shared = multi_gpu.SharedOutputs(path, total, D, create=False)
    out = shared.views(lo, hi)
    ctx = _lib.Context(grp.local_rank % ndev)
    info = ctx.device_info()
    model = _lib.Model(ctx, np.exp(theta), inputs, invQt, invQ, dtype)

    def step():
        model.predict(testing, out=out)
    dt = multi_gpu.timed_steps(grp, step, ctx.synchronize, a.steps, a.warmup)
    # parity of this rank's slice of the gathered array (oracle = checker, after the timed region)
    from oracle import gp_oracle
    idx = np.sort(np.random.RandomState(5 + rank).choice(M, 1024, replace=False))
    ref = gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing[idx])
    errs = [maxnorm_err(r, g[idx]) for r, g in zip(ref, out)]
    tol = 1e-10 if a.precision == "f64" else 1e-4
    worst = grp.max(max(errs))
    if not worst <= tol and not a.no_parity:
        raise SystemExit("bench parity check failed on rank %d: %s" % (rank, errs))
    outd = None
    if rank == 0:
        value = a.steps * total / dt
        outd = {
            "metric": "test-points/sec for predict(mean+var+grad), N_train=300 D=11",
            "value": value, "unit": "test-points/s", "n_gpus": n_gpus_used, "ranks": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: N_train=300, D=11, N_test=%d in all, row-sharded over %d "
                                   "rank(s) (%d rows each), host numpy arrays in, one shared host array out "
                                   "(host gather, no collective), predict mean+var+grad" % (total, world, M),
                       "n_train": N, "n_inputs": D, "n_test_total": total, "n_test_per_gpu": M,
                       "parallelism": "row-sharded x%d, host gather, no collective" % world,
                       "device": info["name"], "host_threads_per_rank": ctx.host_threads(),
                       "cpus_bound_near_gpu": len(near)},
            "roofline": {"bound": "hbm", "achieved": value * 192 / 1e9, "peak": PEAK_HBM_GBPS * n_gpus_used, "unit": "GB/s",
                         "frac": value * 192 / 1e9 / (PEAK_HBM_GBPS * n_gpus_used), "traffic": None,
                         "kernel": "predict_kernel<%s,11,75> behind the host slab pipeline" % ("double" if a.precision == "f64" else "float"),
                         "pcie_gbps_per_gpu": value * 192 / 1e9 / n_gpus_used,
                         "note": "host-to-host: bound by PCIe (192 B per point over a full-duplex link: 57 GB/s one way, "
                                 "2 x 48 GB/s both ways at once, profiles/r03_pcie_duplex.txt; the staged pipeline reaches "
                                 "~65 GB/s for both directions together) and by host DRAM when several GPUs stream at once, "
                                 "not by HBM or the fp64 pipe"},
            "parity": {"worst_over_ranks": worst, "tol": tol, "checked_rows_per_rank": 1024},
        }
        print(json.dumps(outd), flush=True)
    grp.barrier()
    del out
    model.close()
    shared.close(unlink=rank == 0)
    grp.close()
    return outd


def launch_ranks(n_ranks):
    """``python bench.py --gpus N`` with N > 1 and no launcher: start N child ranks of this same
    command line, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run
    sets them), and exit with their status.  This process makes no GPU call before or after the
    children start (they are ordinary child processes, never an exec of a process that has touched
    the GPU); rank 0's JSON line reaches stdout through the inherited pipe.  A rank that fails
    takes the others down with it, so the job never hangs in a barrier and never reports fewer
    GPUs than were asked for."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for pr in list(live):
            code = pr.poll()
            if code is None:
                continue
            live.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in live:          # the exact children started above, nothing else
                    other.terminate()
    if rc:
        sys.stderr.write("bench.py: a rank of the %d-rank job failed (exit %d)\n" % (n_ranks, rc))
    sys.exit(rc)


def devices_for(world):
    """HIP devices visible, and how many of them a ``world``-rank job uses.  A rank needs a device of its own:
    fewer devices than ranks is an error, unless GP_BENCH_SHARE_GPU=1 (rehearsals of the rank plumbing on a
    one-GPU box), and then the line reports the devices really used -- ``n_gpus`` never names more GPUs than ran."""
    from gp_emulator_amd import _lib
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (no HIP device visible); there is no CPU path")
    if world > ndev and os.environ.get("GP_BENCH_SHARE_GPU") != "1":
        raise SystemExit("--gpus %d but only %d HIP device(s) visible (GP_BENCH_SHARE_GPU=1 lets ranks share a "
                         "device for a rehearsal; the line then reports the devices really used)" % (world, ndev))
    return ndev, min(world, ndev)


def dry_run(a, grp):
    """GP_BENCH_DRY_RUN=1 (tests/test_multi_gpu_cpu.py only): the rank plumbing of a --gpus N run
    without any device work -- rendezvous, barrier, max-reduce, rank 0's line.  The line says so
    (``data``), carries no rate and is never a measurement."""
    def step():
        time.sleep(0.001 * (1 + grp.rank))
    from gp_emulator_amd import multi_gpu
    dt = multi_gpu.timed_steps(grp, step, lambda: None, a.steps, a.warmup)
    ranks = grp.sum(1)
    if grp.rank == 0:
        print(json.dumps({"metric": "dry run: rank plumbing only", "value": 0.0, "unit": "test-points/s",
                          "n_gpus": grp.world, "ranks_seen": int(ranks), "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
                          "scaling": a.scaling, "vs_baseline": None, "dtype": a.precision,
                          "data": "dry run (GP_BENCH_DRY_RUN=1): no device work, not a measurement",
                          "config": {"workload": a.workload}}), flush=True)
    grp.close()


def main():
    a = parse()
    if a.cpu_worker:
        return cpu_worker(a.cpu_worker)
    if a.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    launched = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and launched == 1 and "RANK" not in os.environ:
        if a.workload in ("mv", "train"):
            raise SystemExit("--workload %s is a one-GPU workload (--gpus 1)" % a.workload)
        return launch_ranks(a.gpus)       # before anything here has touched the GPU
    if launched != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line would report a GPU count that was not asked for"
                         % (a.gpus, launched))
    if a.workload == "mv":
        return bench_reconstruct(a)
    if a.workload == "train":
        return bench_train(a)
    from gp_emulator_amd import _lib, multi_gpu

    grp = multi_gpu.RankGroup()
    rank, world = grp.rank, grp.world
    if os.environ.get("GP_BENCH_DRY_RUN") == "1":
        return dry_run(a, grp)
    if a.scaling == "strong":
        if a.workload != "c4":
            raise SystemExit("--scaling strong is the c4 workload's mode")
        return bench_strong(a, grp)
    # the CPU legs run first: they start worker processes, which must not happen once this
    # process has initialised the GPU
    cpu = None
    if world == 1 and not a.no_cpu_baseline and a.workload == "c2":
        cpu = cpu_baseline(a.cpu_sample, a.cpu_procs)
    ndev, n_gpus_used = devices_for(world)
    # one process per GPU: run on the cpus next to that GPU (numactl --cpunodebind by hand), so the
    # arrays generated below are first-touched on the socket the device's PCIe root belongs to
    near = _lib.bind_near_device(grp.local_rank % ndev)
    _lib.set_default_device(grp.local_rank % ndev)      # gp.predict(is_gpu=True) in the e2e leg
    ctx = _lib.Context(grp.local_rank % ndev)
    info = ctx.device_info()

    N, D, m_default, kind = WORKLOADS[a.workload]
    dtype = np.float64 if a.precision == "f64" else np.float32
    isz = np.dtype(dtype).itemsize
    M = a.n_test or m_default
    E = a.emulators if kind == "batch" else 1
    inputs, testing, theta, invQ, invQt = synthetic_inputs(1000 + rank, N, D, M)
    d_t = ctx.to_device(testing.astype(dtype))
    if kind == "batch":
        # shared inputs / test rows; per-emulator theta, invQ, invQt from seed + e
        def params(e):
            r = np.random.RandomState(5000 + e)
            return r.random_sample(D + 2), r.random_sample((N, N)), r.random_sample(N)
        thetas = np.empty((E, D + 2))
        invQs = np.empty((E, N, N))
        invQts = np.empty((E, N))
        for e in range(E):
            thetas[e], invQs[e], invQts[e] = params(e)
        model = _lib.BatchModel(ctx, np.exp(thetas), inputs, invQts, invQs, dtype)
        del invQs
    else:
        def params(e):
            return theta, invQ, invQt
        model = _lib.Model(ctx, np.exp(theta), inputs, invQt, invQ, dtype)
    if kind == "hessian":
        d_h = ctx.malloc(M * D * D * isz)
        bufs, outs = [d_t, d_h], [d_h]

        def step():
            model.hessian_device(d_t, d_h, M)
    else:
        d_mu, d_var = ctx.malloc(E * M * isz), ctx.malloc(E * M * isz)
        d_der = ctx.malloc(E * M * D * isz)
        bufs, outs = [d_t, d_mu, d_var, d_der], [d_mu, d_var, d_der]

        def step():
            model.predict_device(d_t, d_mu, d_var, d_der, M, _lib.GP_DERIV_ROWMAJOR)

    e2e = None
    if a.workload == "c2" and not a.no_e2e:
        # host numpy in -> host numpy out through the same library (its own leg, outside the timed
        # region and in front of it: the device is then at working clocks when the resident loop
        # starts, as it is in any run longer than a few tens of milliseconds); on every rank at
        # once when there are several, so the line carries host-to-host scaling too
        e2e = e2e_leg(N, D, M, inputs, testing, theta, invQ, invQt, dtype, grp)

    for _ in range(a.warmup):
        step()
    ctx.synchronize()

    # ---- timed region: exactly K steps, barrier + device sync on both sides ----------
    evs = [ctx.event() for _ in range(a.steps + 1)]
    grp.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.record(evs[0])
    for k in range(a.steps):
        step()
        ctx.record(evs[k + 1])      # HIP events on the stream the kernel is launched on
    ctx.synchronize()
    grp.barrier()
    dt = time.perf_counter() - t0
    dt = grp.max(dt)

    kern_ms = [ctx.elapsed_ms(evs[k], evs[k + 1]) for k in range(a.steps)]
    kern_avg_s = float(np.mean(kern_ms)) * 1e-3

    # parity spot check of what was timed (after the timed region; the oracle enters only here,
    # as the checker, and in the cpu_baseline leg)
    tol = 1e-10 if a.precision == "f64" else 1e-4
    errs, n_checked = sample_parity(ctx, a, kind, outs, inputs, testing, params, M, D, E, dtype)
    if not max(errs.values()) <= tol and not a.no_parity:
        raise SystemExit("bench parity check failed: %s" % errs)

    out = None
    if rank == 0:
        units = E * M                                  # (emulator, test point) pairs per step
        value = world * a.steps * units / dt
        minfo = model.info()
        hess_mfma = (kind == "hessian" and minfo["kernel_d"] in (8, 10, 11, 12, 16)
                     and minfo["kernel_nb"] > 0 and not int(os.environ.get("GP_HESS_VALU", "0")))
        flop_pt = flop_per_point(N, D, kind)
        hess_win = False
        if hess_mfma:       # the windowed form (gp_hessian_win_kernel.hpp) unless GP_HESS_WIN=0
            hess_win = os.environ.get("GP_HESS_WIN", "1") != "0"
        exec_flop_pt = executed_flop_per_point(N, D, kind, minfo, "win" if hess_win else hess_mfma)
        if kind == "hessian":
            byte_pt = (D + D * D) * isz
        else:
            byte_pt = (2 * D + 2) * isz if E == 1 else (2 + D) * isz
        peak = PEAK_FP64_TFLOPS if a.precision == "f64" else 157.3
        executed_tf = exec_flop_pt * units / kern_avg_s / 1e12
        hbm_gbps = byte_pt * units / kern_avg_s / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (a.workload, a.precision))
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = ("profiles/traffic_%s_%s.json (static: rocprofv3 --pmc passes of %s, "
                                  "not measured in this run)" % (a.workload, a.precision, tj.get("round", "an earlier round")))
            except Exception:
                traffic = None
        ctype = "double" if a.precision == "f64" else "float"
        if hess_mfma:
            kfull = "%s<%s,%d,%d>" % ("hessian_win_kernel" if hess_win else "hessian_mfma_kernel", ctype,
                                      minfo["kernel_d"], minfo["kernel_nb"])
        elif kind == "hessian":
            kfull = "hessian_kernel<%s,%d>" % (ctype, minfo["kernel_d"])
        else:
            kfull = "predict_kernel<%s,%d,%d>" % (ctype, minfo["kernel_d"], minfo["kernel_nk"])
        metric = ("test-points/sec for predict(mean+var+grad), N_train=%d D=%d" % (N, D)
                  if kind != "hessian" else
                  "test-points/sec for hessian (full DxD), N_train=%d D=%d" % (N, D))
        desc = {
            "c2": "PROSAIL single-band (BASELINE configs[1]): N_train=250, D=11, N_test=%d per GPU per step, predict mean+var+grad, inputs and outputs resident in HBM" % M,
            "c3": "MultivariateEmulator per-band (BASELINE configs[2]): %d emulators x N_train=250 x D=11, N_test=%d shared test rows in HBM, one launch per step; unit = (emulator, test point)" % (E, M),
            "c4": "BASELINE configs[3]: N_train=300, D=11, N_test=%d per GPU per step (1e8 rows row-sharded over 8 GPUs = 1.25e7 each), predict mean+var+grad" % M,
            "c5": "Hessian path (BASELINE configs[4]): N_train=300, D=16, N_test=%d per GPU per step, full DxD Hessian per test point" % M,
        }[a.workload]
        roof = {"bound": "mfma", "achieved": executed_tf, "peak": peak,
                "unit": "TFLOP/s", "frac": executed_tf / peak, "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": kfull, "kernel_ms": kern_avg_s * 1e3,
                "executed_flop_per_point": exec_flop_pt,
                "flop_per_point": flop_pt,
                "algorithmic_tflops": flop_pt * units / kern_avg_s / 1e12,
                "algorithmic_ratio": flop_pt * units / kern_avg_s / 1e12 / peak,
                "hbm_gbps": hbm_gbps, "hbm_frac": hbm_gbps / PEAK_HBM_GBPS,
                "bytes_per_point": byte_pt,
                "note": ("bound = the fp64 pipe (matrix = vector = 78.6 TFLOP/s datasheet; they share one "
                         "pipe on MI355X, profiles/r01_mfma_f64_probe.txt).  achieved/frac = flops EXECUTED "
                         "(matrix instructions issued after symmetric block-pair folding + phase-A "
                         "arithmetic) x units / HIP-event kernel time; algorithmic_ratio = SURVEY.md 8d's "
                         "un-halved count over the same peak (it exceeds 1 because folding halves the "
                         "variance work); hbm_frac = algorithmic bytes against 8 TB/s")}
        out = {
            "metric": metric,
            "value": value, "unit": "test-points/s", "n_gpus": n_gpus_used, "ranks": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": desc, "n_train": N, "n_inputs": D, "n_test_per_gpu": M,
                       "n_emulators": E,
                       "parallelism": "row-sharded x%d, no collective" % world,
                       "device": info["name"], "compute_units": info["compute_units"],
                       "cpus_bound_near_gpu": len(near)},
            "roofline": roof,
            "parity": dict(errs, tol=tol, checked_rows=int(n_checked)),
        }
    if rank == 0:
        if e2e is not None:
            rate, steady, first, pinned_s = e2e
            roof["e2e_points_per_s"] = rate
            roof["e2e_ms_per_call"] = steady * 1e3
            roof["e2e_first_call_ms"] = first * 1e3
            out["end_to_end"] = {"value": rate, "unit": "test-points/s", "ms_per_call": steady * 1e3,
                                 "first_call_ms": first * 1e3, "rows_per_gpu": M, "n_gpus": n_gpus_used, "threshold": 2e5,
                                 "what": "gp.predict(testing, is_gpu=True) from host numpy arrays to host numpy "
                                         "arrays, median of 10 calls after 3 warm-up calls (PCIe and host copies "
                                         "included; tests/benchmark.py:41-44); the result arrays are pooled buffers "
                                         "(_lib.OutputPool: each call's results are dropped before the next call, so "
                                         "their memory is reused; a caller that KEEPS every result pays ~7 ms of page "
                                         "faults per 1e6 rows on top, profiles/r02_host_path_experiments.txt); with "
                                         "several ranks all of them call at once and a call lasts as long as its "
                                         "slowest rank"}
            if pinned_s:
                roof["e2e_pinned_points_per_s"] = world * M / pinned_s
                out["end_to_end"]["pinned_arrays_value"] = world * M / pinned_s
                out["end_to_end"]["pinned_arrays_ms_per_call"] = pinned_s * 1e3
                out["end_to_end"]["pinned_arrays_what"] = ("the same call with test rows and out= arrays from "
                                                           "gp_emulator_amd.pinned_empty: no staging, no host copies")
            out["warmup_extra"] = ("the 13 + 8 host-to-host calls of end_to_end run in front of the --warmup steps (about 45 ms "
                                   "of device work): the timed region starts at working clocks")
        if cpu is not None:
            out["cpu_baseline"] = cpu
            cpu["gpu_over_cpu"] = value / cpu["value"]          # resident GPU rate / reference-style CPU
            if "end_to_end" in out:
                cpu["e2e_over_cpu"] = out["end_to_end"]["value"] / cpu["value"]
                cpu["e2e_over_all_cores"] = out["end_to_end"]["value"] / cpu["all_cores_value"]
        print(json.dumps(out), flush=True)

    for p in bufs:
        ctx.free(p)
    for e in evs:
        ctx.event_destroy(e)
    grp.close()
    return out


if __name__ == "__main__":
    main()
