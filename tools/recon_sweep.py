"""Reconstruction kernel: achieved HBM write rate vs band count / alignment / row count."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import _lib
ctx = _lib.Context(0)
rs = np.random.RandomState(0)
P = 12
for dt in (np.float64, np.float32):
    isz = np.dtype(dt).itemsize
    for B, R in ((2101, 1000000), (2104, 1000000), (2112, 1000000), (2048, 1000000), (2101, 100000)):
        basis = rs.standard_normal((P, B)).astype(dt)
        coef = rs.standard_normal((P, R)).astype(dt)
        d_b, d_c = ctx.to_device(basis), ctx.to_device(coef)
        d_o = ctx.malloc(R * B * isz)
        for _ in range(2):
            ctx.reconstruct_device(dt, d_b, d_c, d_o, R, P, B)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(3):
            ctx.reconstruct_device(dt, d_b, d_c, d_o, R, P, B)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 3
        print("%s B=%d R=%d: %.3f ms  %.0f GB/s" % (np.dtype(dt).name, B, R, ms, R * B * isz / ms / 1e6), flush=True)
        for p in (d_b, d_c, d_o):
            ctx.free(p)
