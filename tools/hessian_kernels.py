#!/usr/bin/env python3
"""Hessian kernels by dimension: HIP-event time per launch of Model.hessian_device on 1e6 resident
test rows, both precisions.  Run twice to compare the two kernels:

    python tools/hessian_kernels.py                  # matrix-core kernel for kernel D >= 8
    GP_HESS_VALU=1 python tools/hessian_kernels.py   # VALU kernel for every D
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import _lib  # noqa: E402
from bench import synthetic_inputs  # noqa: E402  (seeded synthetic inputs)

ctx = _lib.default_context(0)
M = 1000000
print("GP_HESS_VALU=%s" % os.environ.get("GP_HESS_VALU", "0"))
for N, D in ((250, 8), (250, 10), (250, 11), (300, 12), (300, 16), (120, 8)):
    inputs, testing, theta, invQ, invQt = synthetic_inputs(1, N, D, M)
    for prec in (np.float64, np.float32):
        m = _lib.Model(ctx, np.exp(theta), inputs, invQt, None, prec)
        d_t = ctx.to_device(np.ascontiguousarray(testing, dtype=prec))
        d_h = ctx.malloc(M * D * D * np.dtype(prec).itemsize)
        for _ in range(2):
            m.hessian_device(d_t, d_h, M)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(5):
            m.hessian_device(d_t, d_h, M)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 5
        print("N=%d D=%d %s: %.3f ms  %.3g rows/s" % (N, D, np.dtype(prec).name, ms, M / ms * 1e3), flush=True)
        ctx.free(d_t)
        ctx.free(d_h)
        m.close()
