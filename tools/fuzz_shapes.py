#!/usr/bin/env python3
"""Random shapes through the drop-in API against its own numpy branch: n_train 1..330, n_inputs 1..17,
1..3000 rows, both precisions, predict (both boundary flows) and Hessian, both predict kernel forms.
A soak for edge cases the parametrised tests do not enumerate; prints the worst errors.

    python tools/fuzz_shapes.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_emulator_amd import GaussianProcess  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"f64": 0.0, "f32": 0.0, "h64": 0.0, "h32": 0.0}


def err(ref, got):
    return float(np.max(np.abs(ref - got)) / max(np.max(np.abs(ref)), 1e-300))


for case in range(n_cases):
    N = int(rs.choice([1, 2, 3, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 191, 193, 240, 241, 250, 255, 256,
                       257, 300, 304, 305, 319, 320, 321, 330, rs.randint(1, 331)]))
    D = int(rs.randint(1, 18))
    M = int(rs.choice([1, 2, 15, 16, 17, 63, 65, 127, 129, 1000, rs.randint(1, 3001), 8191, 8193, 20000]))
    inputs, testing = rs.random_sample((N, D)), rs.random_sample((M, D))
    theta, invQ, invQt = rs.random_sample(D + 2), rs.random_sample((N, N)), rs.random_sample(N)
    gp = GaussianProcess(inputs, [])
    gp.theta, gp.invQ, gp.invQt = theta, invQ, invQt
    ref = gp.predict(testing)
    os.environ["GP_NO_FEW"] = str(case % 2)
    gp.row_major_boundary = bool(rs.randint(2))
    for prec, key, tol in ((np.float64, "f64", 1e-10), (np.float32, "f32", 1e-4)):
        got = gp.predict(testing, is_gpu=True, precision=prec, threshold=float(rs.choice([2e5, 700, 5000])))
        e = max(err(r, g) for r, g in zip(ref, got))
        worst[key] = max(worst[key], e)
        assert e <= tol, ("predict", case, N, D, M, prec, e)
    if D <= 16:
        mh = min(M, 200)
        href = gp.hessian(testing[:mh])
        for prec, key, tol in ((np.float64, "h64", 1e-10), (np.float32, "h32", 1e-4)):
            e = err(href, gp.hessian(testing[:mh], is_gpu=True, precision=prec))
            worst[key] = max(worst[key], e)
            assert e <= tol, ("hessian", case, N, D, mh, prec, e)
# batched emulators on shared inputs (perband.predict_bands; both kernel forms, emulator shards on one device)
from gp_emulator_amd import perband  # noqa: E402
n_batched = max(1, n_cases // 5)
worst_b = 0.0
for case in range(n_batched):
    N, D = int(rs.randint(1, 321)), int(rs.randint(1, 17))
    E, M = int(rs.randint(2, 10)), int(rs.choice([1, 16, 17, 100, 777, 3000]))
    inputs, testing = rs.random_sample((N, D)), rs.random_sample((M, D))
    gps = []
    for e in range(E):
        gp = GaussianProcess(inputs, [])
        gp.theta, gp.invQ, gp.invQt = rs.random_sample(D + 2), rs.random_sample((N, N)), rs.random_sample(N)
        gps.append(gp)
    os.environ["GP_NO_FEW"] = str(case % 2)
    devs = None if case % 3 else [0] * int(rs.randint(1, 4))
    got = perband.predict_bands(gps, testing, devices=devs)
    for e, gp in enumerate(gps):
        ref = gp.predict(testing)
        err_e = max(err(r, g[e]) for r, g in zip(ref, got))
        worst_b = max(worst_b, err_e)
        assert err_e <= 1e-10, ("batched", case, N, D, E, M, e, err_e)
print("batched OK: %d cases, worst error %.2e" % (n_batched, worst_b))
print("fuzz OK: %d cases; worst errors predict fp64 %.2e fp32 %.2e, Hessian fp64 %.2e fp32 %.2e" % (
    n_cases, worst["f64"], worst["f32"], worst["h64"], worst["h32"]))
