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

Extra objects on the JSON line:
  roofline      the fused kernel against the FP64 matrix-core peak (the binding roofline,
                arithmetic intensity ~746 flop/B); ``hbm`` gives the same launch against
                the HBM roofline, which is what BASELINE.json's north star asks to see.
  cpu_baseline  the numpy path (oracle = py3 restatement of the reference's cpu_predict)
                timed on this box's host cores on a bounded sample; rank 0, N=1 only.
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
    return ap.parse_args()


def cpu_baseline(sample_rows):
    """numpy path on the host: the oracle's restatement of GaussianProcess.cpu_predict
    (scipy cdist + numpy; BLAS threads = all cores for the np.dot calls, everything else
    single-threaded -- exactly how the reference runs it)."""
    from oracle import gp_oracle
    inputs, testing, theta, invQ, invQt = synthetic_inputs(12345, N_TRAIN, N_INPUTS, sample_rows)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    from threadpoolctl import threadpool_limits
    blas_threads = max(1, min(avail, 16))       # the GPU box gives one GPU a 16-cpu share
    with threadpool_limits(limits=blas_threads):
        gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing[:2000])   # warm up
        t0 = time.perf_counter()
        gp_oracle.cpu_predict_blocked(inputs, theta, invQ, invQt, testing, block=50000)
        dt = time.perf_counter() - t0
    return {"value": sample_rows / dt, "unit": "test-points/s",
            "cores": int(blas_threads), "kind": "port",
            "sample": "%d rows of the same N=250, D=11 workload in 50k-row blocks, %.1f s; "
                      "numpy+scipy path (oracle/gp_oracle.py), BLAS threads=%s of %d host "
                      "cpus, elementwise/cdist/exp single-threaded as in the reference"
                      % (sample_rows, dt, blas_threads, os.cpu_count() or 0)}


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
    n_pass = (N + 7) // 8 + 2
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
           "roofline": {"bound": "hbm", "achieved": E * n_pass * 2 * N * N * 8 / dt / 1e9, "peak": PEAK_HBM_GBPS,
                        "unit": "GB/s", "frac": E * n_pass * 2 * N * N * 8 / dt / 1e9 / PEAK_HBM_GBPS,
                        "traffic": None, "kernel": "likelihood_kernel",
                        "note": "Gauss-Jordan, 8 pivots per pass: the N x N workspace is read and written "
                                "ceil(N/8) + 2 times per theta (build, passes, gradient), 2 N^2 x 8 B each, "
                                "mostly from the Infinity Cache; achieved = those bytes / wall time "
                                "(host copies included)"},
           "cpu_baseline": {"value": 1.0 / cpu_per, "unit": "theta-evaluations/s", "cores": int(os.cpu_count() or 1),
                            "kind": "port", "sample": "%d evaluations of the numpy path (oracle)" % n_cpu},
           "parity": {"checked_sets": n_cpu, "tol_cost": 1e-8, "tol_grad": 1e-5}}
    print(json.dumps(out), flush=True)
    return out


def main():
    a = parse()
    if a.workload == "mv":
        return bench_reconstruct(a)
    if a.workload == "train":
        return bench_train(a)
    from gp_emulator_amd import _lib, multi_gpu

    grp = multi_gpu.RankGroup()
    rank, world = grp.rank, grp.world
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (no HIP device visible); there is no CPU path")
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
        thetas = np.empty((E, D + 2))
        invQs = np.empty((E, N, N), dtype=dtype)
        invQts = np.empty((E, N))
        for e in range(E):
            r = np.random.RandomState(5000 + e)
            thetas[e] = r.random_sample(D + 2)
            invQs[e] = r.random_sample((N, N))
            invQts[e] = r.random_sample(N)
        model = _lib.BatchModel(ctx, np.exp(thetas), inputs, invQts, invQs, dtype)
        if E > 64:
            del invQs
    else:
        model = _lib.Model(ctx, np.exp(theta), inputs, invQt, invQ, dtype)
    if kind == "hessian":
        d_h = ctx.malloc(M * D * D * isz)
        bufs = [d_t, d_h]

        def step():
            model.hessian_device(d_t, d_h, M)
    else:
        d_mu, d_var = ctx.malloc(E * M * isz), ctx.malloc(E * M * isz)
        d_der = ctx.malloc(E * M * D * isz)
        bufs = [d_t, d_mu, d_var, d_der]

        def step():
            model.predict_device(d_t, d_mu, d_var, d_der, M, _lib.GP_DERIV_ROWMAJOR)

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

    # parity spot check of what was timed (after the timed region: the numpy check leaves
    # BLAS worker threads spinning, which would steal host time from the launch loop)
    # (the oracle enters only here, as the checker, and in the cpu_baseline leg)
    from oracle import gp_oracle
    rs = np.random.RandomState(5)
    tol = 1e-10 if a.precision == "f64" else 1e-4
    if kind == "hessian":
        idx = np.sort(rs.choice(M, 256, replace=False))
        got = ctx.to_host(d_h, (M, D, D), dtype)[idx]
        ref = gp_oracle.hessian(inputs, theta, invQt, testing[idx])
        errs = [maxnorm_err(ref, got)]
        names = ["e_hess"]
    else:
        idx = rs.choice(M, 2048, replace=False)
        e_chk = E - 1
        mu = ctx.to_host(d_mu, (E, M), dtype)[e_chk, idx]
        var = ctx.to_host(d_var, (E, M), dtype)[e_chk, idx]
        if E * M * D * isz < (4 << 30):
            der = ctx.to_host(d_der, (E, M, D), dtype)[e_chk, idx]
        else:                             # config 3 at full size: 18 GB of gradients
            der = None
        if kind == "batch":
            r = np.random.RandomState(5000 + e_chk)
            th_c, iq_c, iqt_c = r.random_sample(D + 2), r.random_sample((N, N)), r.random_sample(N)
        else:
            th_c, iq_c, iqt_c = theta, invQ, invQt
        ref = gp_oracle.cpu_predict(inputs, th_c, iq_c, iqt_c, testing[idx])
        errs = [maxnorm_err(ref[0], mu), maxnorm_err(ref[1], var)]
        names = ["e_mu", "e_var"]
        if der is not None:
            errs.append(maxnorm_err(ref[2], der))
            names.append("e_deriv")
    if not max(errs) <= tol and not a.no_parity:
        raise SystemExit("bench parity check failed: %s" % dict(zip(names, errs)))

    out = None
    if rank == 0:
        units = E * M                                  # (emulator, test point) pairs per step
        value = world * a.steps * units / dt
        flop_pt = flop_per_point(N, D, kind)
        if kind == "hessian":
            byte_pt = (D + D * D) * isz
        else:
            byte_pt = (2 * D + 2) * isz if E == 1 else (2 + D) * isz
        peak = PEAK_FP64_TFLOPS if a.precision == "f64" else 157.3
        achieved_tf = flop_pt * units / kern_avg_s / 1e12
        hbm_gbps = byte_pt * units / kern_avg_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s_%s.json" % (a.workload, a.precision))
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # what the kernel really executes on the dominant pipe (not the algorithmic count above):
        # folded variance = 4 NB(NB+1)/2 MFMAs of 16x16x4 per 16-row tile; Hessian = D(D+1)/2 fmas
        # per (training point, test point) over the padded training set
        nb_pad = model.info()["kernel_nb"] if kind != "hessian" else (N + 15) // 16
        hess_mfma = (kind == "hessian" and model.info()["kernel_d"] in (8, 10, 11, 12, 16)
                     and model.info()["kernel_nb"] > 0 and not int(os.environ.get("GP_HESS_VALU", "0")))
        if hess_mfma:
            # 4 x 4 blocks (bi <= bj) of the D x D matrix, 4 NB MFMAs of 16x16x4 each per 16-row tile
            nb4 = (model.info()["kernel_d"] + 3) // 4
            exec_flop_pt = nb4 * (nb4 + 1) // 2 * 4 * model.info()["kernel_nb"] * 2048 // 16
        elif kind == "hessian":
            kd = model.info()["kernel_d"]
            exec_flop_pt = 2 * 16 * nb_pad * kd * (kd + 1) // 2
        else:
            exec_flop_pt = 4 * nb_pad * (nb_pad + 1) // 2 * 2048 // 16
        executed_tf = exec_flop_pt * units / kern_avg_s / 1e12
        kname = {"predict": "predict_kernel", "batch": "predict_kernel", "hessian": "hessian_kernel"}[kind]
        ctype = "double" if a.precision == "f64" else "float"
        minfo = model.info()
        if hess_mfma:
            kname = "hessian_mfma_kernel"
        kfull = ("%s<%s,%d,%d>" % (kname, ctype, minfo["kernel_d"], minfo["kernel_nb"])
                 if kind != "hessian" or hess_mfma else "%s<%s,%d>" % (kname, ctype, minfo["kernel_d"]))
        metric = ("test-points/sec for predict(mean+var+grad), N_train=%d D=%d" % (N, D)
                  if kind != "hessian" else
                  "test-points/sec for hessian (full DxD), N_train=%d D=%d" % (N, D))
        desc = {
            "c2": "PROSAIL single-band (BASELINE configs[1]): N_train=250, D=11, N_test=%d per GPU per step, predict mean+var+grad, inputs and outputs resident in HBM" % M,
            "c3": "MultivariateEmulator per-band (BASELINE configs[2]): %d emulators x N_train=250 x D=11, N_test=%d shared test rows in HBM, one launch per step; unit = (emulator, test point)" % (E, M),
            "c4": "BASELINE configs[3]: N_train=300, D=11, N_test=%d per GPU per step (1e8 rows row-sharded over 8 GPUs = 1.25e7 each), predict mean+var+grad" % M,
            "c5": "Hessian path (BASELINE configs[4]): N_train=300, D=16, N_test=%d per GPU per step, full DxD Hessian per test point" % M,
        }[a.workload]
        out = {
            "metric": metric,
            "value": value, "unit": "test-points/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": desc, "n_train": N, "n_inputs": D, "n_test_per_gpu": M,
                       "n_emulators": E,
                       "parallelism": "row-sharded x%d, no collective" % world,
                       "device": info["name"], "compute_units": info["compute_units"]},
            "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved_tf / peak, "traffic": traffic,
                         "kernel": kfull, "kernel_ms": kern_avg_s * 1e3,
                         "flop_per_point": flop_pt,
                         "executed": {"flop_per_point": exec_flop_pt, "achieved": executed_tf,
                                      "frac": executed_tf / peak,
                                      "what": ("variance MFMAs actually issued (symmetric folding)"
                                               if kind != "hessian" else
                                               "pair-product MFMAs actually issued (4 x 4 blocks)" if hess_mfma
                                               else "pair-product fmas actually issued (upper triangle)")},
                         "note": ("achieved = algorithmic flop/pt of SURVEY.md 8d (un-halved variance "
                                  "contraction / Hessian as the reference writes it) x units per launch "
                                  "/ HIP-event kernel time.  The kernel EXECUTES fewer flops than that "
                                  "(symmetric block-pair folding of the variance: 0.53x of its MFMAs; "
                                  "Hessian: D(D+1)/2 products), which is how frac can exceed 1; peak is "
                                  "the fp64 matrix (= vector) datasheet rate, measured MFMA-only ceiling "
                                  "on this chip 71.4 TFLOP/s (profiles/r01_mfma_f64_probe.txt)"),
                         "hbm": {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                 "frac": hbm_gbps / PEAK_HBM_GBPS, "bytes_per_point": byte_pt}},
            "parity": dict(zip(names, errs), tol=tol, checked_rows=int(len(idx))),
        }
        if world == 1 and not a.no_cpu_baseline and a.workload == "c2":
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    for p in bufs:
        ctx.free(p)
    for e in evs:
        ctx.event_destroy(e)
    grp.close()
    return out


if __name__ == "__main__":
    main()
