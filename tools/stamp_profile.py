"""Where does an item of the predict kernel spend its cycles?  Needs a library built with
-DGP_STAMPS=1 (GP_PREDICT_LIB=...): every wave sums the shader cycles of six segments of its
items; shares are printed.  The stamped build's run time is never quoted."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import _lib
from bench import synthetic_inputs

ctx = _lib.Context(0)
DT = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
ISZ = np.dtype(DT).itemsize
WAVES = 12 if DT == np.float32 else 8
M = 1000000
inputs, testing, theta, invQ, invQt = synthetic_inputs(1, 250, 11, M)
model = _lib.Model(ctx, np.exp(theta), inputs, invQt, invQ, DT)
d_t = ctx.to_device(testing.astype(DT))
d_mu, d_var, d_der = ctx.malloc(M * ISZ), ctx.malloc(M * ISZ), ctx.malloc(M * 11 * ISZ)
d_dbg = ctx.to_device(np.zeros(8, np.uint64))
_lib.check(ctx.lib.gp_ctx_set_debug_buffer(ctx.h, d_dbg))
for _ in range(2):
    model.predict_device(d_t, d_mu, d_var, d_der, M)
ctx.synchronize()
ctx.h2d(d_dbg, np.zeros(8, np.uint64))
K = 5
for _ in range(K):
    model.predict_device(d_t, d_mu, d_var, d_der, M)
ctx.synchronize()
s = ctx.to_host(d_dbg, (8,), np.uint64).astype(np.float64)
waves = s[7] / K
names = ["0 emulator switch + barrier + DMA issue", "1 test rows load + scale", "2 phase A (kernel row, mean, grad sums)",
         "3 lane-group reductions + mu/deriv stores", "4 phase B (MFMA variance)", "5 var reduction + store"]
tot = s[:6].sum()
items_per_wave = (M / (16 * WAVES)) / (waves / WAVES)
print("waves per launch %.0f, items per wave %.2f" % (waves, items_per_wave))
for n, v in zip(names, s[:6]):
    print("%-46s %5.1f %%   %8.0f cycles per item" % (n, 100 * v / tot, v / K / waves / items_per_wave))
print("total cycles per item (per wave) %.0f" % (tot / K / waves / items_per_wave))
