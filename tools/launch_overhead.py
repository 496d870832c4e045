"""Diagnostic: where does host wall time go around back-to-back launches?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import _lib
from bench import synthetic_inputs

ctx = _lib.Context(0)
M = 1000000
inputs, testing, theta, invQ, invQt = synthetic_inputs(1, 250, 11, M)
model = _lib.Model(ctx, np.exp(theta), inputs, invQt, invQ, np.float64)
d_t = ctx.to_device(testing)
d_mu, d_var, d_der = ctx.malloc(M * 8), ctx.malloc(M * 8), ctx.malloc(M * 88)
for _ in range(3):
    model.predict_device(d_t, d_mu, d_var, d_der, M)
ctx.synchronize()
for K in (1, 10, 30, 100):
    t0 = time.perf_counter()
    for _ in range(K):
        model.predict_device(d_t, d_mu, d_var, d_der, M)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print("K=%3d no events: enqueue %.3f ms, total %.3f ms, per step %.3f ms" % (K, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / K))
for K in (10, 30):
    evs = [ctx.event() for _ in range(K + 1)]
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.record(evs[0])
    for k in range(K):
        model.predict_device(d_t, d_mu, d_var, d_der, M)
        ctx.record(evs[k + 1])
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    ms = [ctx.elapsed_ms(evs[k], evs[k + 1]) for k in range(K)]
    print("K=%3d events: enqueue %.3f ms, total %.3f ms, per step %.3f ms; event sum %.3f, first %.3f min %.3f max %.3f" % (
        K, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / K, sum(ms), ms[0], min(ms), max(ms)))
    tot = ctx.elapsed_ms(evs[0], evs[K])
    print("      elapsed(first,last) = %.3f ms" % tot)
