"""End-to-end (host numpy arrays in -> host numpy arrays out) timing of the drop-in path,
i.e. what tests/benchmark.py:41-44 times, next to the kernel-only rate.

    python tools/host_path_timing.py [--c4]

Environment knobs read by the library: GP_HOST_THREADS (default 8), GP_HOST_HUGEPAGES (1).
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import GaussianProcess, _lib, multi_gpu
from bench import synthetic_inputs


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), float(np.median(ts))


def main():
    if "--bind" in sys.argv:       # run next to the GPU, as bench.py does
        print("bound to %d cpus next to device 0" % len(_lib.bind_near_device(0)))
    ctx = _lib.default_context(0)
    print("host threads = %d, GP_HOST_HUGEPAGES=%s, THP=%s" % (
        ctx.host_threads(), os.environ.get("GP_HOST_HUGEPAGES", "1"),
        open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
        if os.path.exists("/sys/kernel/mm/transparent_hugepage/enabled") else "?"), flush=True)
    N, D = 250, 11
    quick = "--quick" in sys.argv
    for M in ((1000000,) if quick else (100000, 1000000)):
        inputs, testing, theta, invQ, invQt = synthetic_inputs(1, N, D, M)
        gp = GaussianProcess(inputs, [])
        gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
        for prec in ((np.float64,) if quick else (np.float64, np.float32)):
            name = np.dtype(prec).name
            for thr in ((2e5,) if quick else (2e5, 1e7)):
                gp.predict(testing[:1000], is_gpu=True, precision=prec, threshold=thr)
                lo, med = best(lambda: gp.predict(testing, is_gpu=True, precision=prec, threshold=thr))
                print("M=%d %s threshold=%g: predict(is_gpu=True), fresh outputs: %.2f ms (median %.2f) -> %.3g pts/s"
                      % (M, name, thr, lo * 1e3, med * 1e3, M / lo), flush=True)
            m = gp.gpu_model(prec)
            out = m.predict(testing)
            lo, med = best(lambda: m.predict(testing, out=out))
            print("M=%d %s Model.predict(out=reused arrays): %.2f ms (median %.2f) -> %.3g pts/s"
                  % (M, name, lo * 1e3, med * 1e3, M / lo), flush=True)
        if quick:
            continue
        gp.row_major_boundary = False
        lo, med = best(lambda: gp.predict(testing, is_gpu=True, precision=np.float64, threshold=2e5))
        print("M=%d float64 threshold=2e5 through predict_wrap blocks (reference flow): %.2f ms -> %.3g pts/s"
              % (M, lo * 1e3, M / lo), flush=True)
    if "--c4" in sys.argv:
        # one C4 shard (BASELINE configs[3]: 1e8 rows over 8 GPUs = 1.25e7 rows each, N=300, D=11)
        N, D, M = 300, 11, 12500000
        inputs, testing, theta, invQ, invQt = synthetic_inputs(7, N, D, M)
        gp = GaussianProcess(inputs, [])
        gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
        out = multi_gpu.predict_sharded(gp, testing[:100000], devices=[0])
        t0 = time.perf_counter()
        out = multi_gpu.predict_sharded(gp, testing, devices=[0])
        dt_fresh = time.perf_counter() - t0
        lo, med = best(lambda: multi_gpu.predict_sharded(gp, testing, devices=[0], out=out), reps=3)
        idx = np.random.RandomState(0).choice(M, 2000, replace=False)
        ref = gp.predict(testing[idx], is_gpu=False)          # the API's explicit numpy branch
        errs = [float(np.max(np.abs(r - o[idx])) / np.max(np.abs(r))) for r, o in zip(ref, out)]
        print("C4 shard N=300 D=11 M=%d predict_sharded(devices=[0]): fresh outputs %.1f ms, reused %.1f ms "
              "-> %.3g pts/s; parity vs the numpy branch on 2000 rows: %.2g %.2g %.2g"
              % (M, dt_fresh * 1e3, lo * 1e3, M / lo, *errs), flush=True)


if __name__ == "__main__":
    main()
