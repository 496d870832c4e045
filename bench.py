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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--n-test", type=int, default=N_TEST)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1000000)
    return ap.parse_args()


def cpu_baseline(sample_rows):
    """numpy path on the host: the oracle's restatement of GaussianProcess.cpu_predict
    (scipy cdist + numpy; BLAS threads = all cores for the np.dot calls, everything else
    single-threaded -- exactly how the reference runs it)."""
    from oracle import gp_oracle
    inputs, testing, theta, invQ, invQt = gp_oracle.benchmark_inputs(12345, N_TRAIN, N_INPUTS,
                                                                     sample_rows)
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


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    dist = None
    if world > 1:
        import torch  # plumbing only: rendezvous, barrier, max-reduce over ranks
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from gp_emulator_amd import _lib
    from oracle import gp_oracle  # inputs recipe (and the cpu_baseline leg) only

    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (no HIP device visible); there is no CPU path")
    ctx = _lib.Context(local_rank % ndev)
    info = ctx.device_info()

    dtype = np.float64 if a.precision == "f64" else np.float32
    M = a.n_test
    inputs, testing, theta, invQ, invQt = gp_oracle.benchmark_inputs(1000 + rank, N_TRAIN,
                                                                     N_INPUTS, M)
    model = _lib.Model(ctx, np.exp(theta), inputs, invQt, invQ, dtype)
    isz = np.dtype(dtype).itemsize
    d_t = ctx.to_device(testing.astype(dtype))
    d_mu, d_var = ctx.malloc(M * isz), ctx.malloc(M * isz)
    d_der = ctx.malloc(M * N_INPUTS * isz)

    def step():
        model.predict_device(d_t, d_mu, d_var, d_der, M, _lib.GP_DERIV_ROWMAJOR)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    ctx.synchronize()

    # ---- timed region: exactly K steps, barrier + device sync on both sides ----------
    evs = [ctx.event() for _ in range(a.steps + 1)]
    barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.record(evs[0])
    for k in range(a.steps):
        step()
        ctx.record(evs[k + 1])      # HIP events on the stream the kernel is launched on
    ctx.synchronize()
    barrier()
    dt = time.perf_counter() - t0

    kern_ms = [ctx.elapsed_ms(evs[k], evs[k + 1]) for k in range(a.steps)]
    kern_avg_s = float(np.mean(kern_ms)) * 1e-3

    # parity spot check of what was timed (after the timed region: the numpy check leaves
    # BLAS worker threads spinning, which would steal host time from the launch loop)
    idx = np.random.RandomState(5).choice(M, 2048, replace=False)
    mu = ctx.to_host(d_mu, (M,), dtype)[idx]
    var = ctx.to_host(d_var, (M,), dtype)[idx]
    der = ctx.to_host(d_der, (M, N_INPUTS), dtype)[idx]
    ref = gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing[idx])
    errs = [gp_oracle.maxnorm_err(r, g) for r, g in zip(ref, (mu, var, der))]
    tol = 1e-10 if a.precision == "f64" else 1e-4
    if not max(errs) <= tol:
        raise SystemExit("bench parity check failed: %s" % errs)

    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        value = world * a.steps * M / dt
        flop_pt = FLOP_PER_POINT
        byte_pt = BYTES_PER_POINT if a.precision == "f64" else BYTES_PER_POINT // 2
        peak = PEAK_FP64_TFLOPS if a.precision == "f64" else 157.3
        achieved_tf = flop_pt * M / kern_avg_s / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % a.precision)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "test-points/sec for predict(mean+var+grad), N_train=250 D=11",
            "value": value, "unit": "test-points/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "PROSAIL single-band (BASELINE configs[1]): N_train=250, "
                                   "D=11, N_test=%d per GPU per step, predict mean+var+grad, "
                                   "inputs and outputs resident in HBM" % M,
                       "n_train": N_TRAIN, "n_inputs": N_INPUTS, "n_test_per_gpu": M,
                       "parallelism": "row-sharded x%d, no collective" % world,
                       "device": info["name"], "compute_units": info["compute_units"]},
            "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved_tf / peak, "traffic": traffic,
                         "kernel": "predict_kernel<%s,11,16>" % ("double" if a.precision == "f64" else "float"),
                         "kernel_ms": kern_avg_s * 1e3,
                         "flop_per_point": flop_pt,
                         "note": "achieved = algorithmic 143262 flop/pt (un-halved variance "
                                 "contraction, as the reference computes it) x points per "
                                 "launch / HIP-event kernel time; the kernel executes "
                                 "~0.53x of those MFMA flops (symmetric block-pair folding)",
                         "hbm": {"achieved": byte_pt * M / kern_avg_s / 1e9,
                                 "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                 "frac": byte_pt * M / kern_avg_s / 1e9 / PEAK_HBM_GBPS,
                                 "bytes_per_point": byte_pt}},
            "parity": {"e_mu": errs[0], "e_var": errs[1], "e_deriv": errs[2], "tol": tol,
                       "checked_rows": 2048},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    for p in (d_t, d_mu, d_var, d_der):
        ctx.free(p)
    for e in evs:
        ctx.event_destroy(e)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
