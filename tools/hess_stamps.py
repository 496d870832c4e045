"""Where does an item of hessian_win_kernel spend its cycles?  Needs a library built with
-DGP_STAMPS=1 (GP_PREDICT_LIB=...): every wave sums the shader cycles of five segments of its
items; shares are printed.  The stamped build's run time is never quoted.

    python -m gp_emulator_amd.build --define GP_STAMPS=1 --lib gp_emulator_amd/libgp_predict_hip_stamps.so
    GP_PREDICT_LIB=gp_emulator_amd/libgp_predict_hip_stamps.so python tools/hess_stamps.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import _lib
from bench import synthetic_inputs

ctx = _lib.Context(0)
N, D, M, WAVES = 300, 16, 1000000, 8
inputs, testing, theta, invQ, invQt = synthetic_inputs(1, N, D, M)
model = _lib.Model(ctx, np.exp(theta), inputs, invQt, invQ, np.float64)
d_t = ctx.to_device(testing)
d_h = ctx.malloc(M * D * D * 8)
init = np.zeros(32 + 4 * 512 + 240 * 512, np.uint64); init[8] = init[16] = init[18] = np.iinfo(np.uint64).max
d_dbg = ctx.to_device(init)
_lib.check(ctx.lib.gp_ctx_set_debug_buffer(ctx.h, d_dbg))
for _ in range(2):
    model.hessian_device(d_t, d_h, M)
ctx.synchronize()
ctx.h2d(d_dbg, init)
K = int(os.environ.get('GP_STAMP_LAUNCHES', '5'))
for _ in range(K):
    model.hessian_device(d_t, d_h, M)
ctx.synchronize()
s = ctx.to_host(d_dbg, (32 + 4 * 512 + 240 * 512,), np.uint64).astype(np.float64)
raw = ctx.to_host(d_dbg, (32 + 4 * 512 + 240 * 512,), np.uint64)
waves = s[7] / K
names = ["0 item start: barrier, DMA issue, test row", "1 phase A (all windows)", "2 phase B (matrix instructions, chunk barriers)",
         "3 s / G from their accumulator slots", "4 finish + stores", "5 chunk boundaries: DMA wait + barrier", "6 barrier in front of the whole-line finish"]
tot = s[:7].sum()
items_per_wave = (M / (16 * WAVES)) / (waves / WAVES)
print("N=%d D=%d: waves per launch %.0f, items per wave %.2f" % (N, D, waves, items_per_wave))
for n, v in zip(names, s[:7]):
    print("%-52s %5.1f %%   %8.0f cycles per item" % (n, 100 * v / tot, v / K / waves / items_per_wave))
print("total cycles per item (per wave) %.0f" % (tot / K / waves / items_per_wave))
print("wave lifetime per launch (cycles): shortest %.0f  longest %.0f (of the %d launches)  mean %.0f; waves of the first half of the grid: mean %.0f"
      % (s[8], s[9], K, s[10] / s[7], s[11] / max(s[12], 1)))
print("items per launch: first half of the grid %.0f, second half %.0f" % (s[13] / K, s[14] / K))
if K == 1:
    r0 = int(raw[16])
    print("one launch, 100 MHz ticks from the earliest wave start: latest start %d, earliest end %d, latest end %d"
          % (int(raw[17]) - r0, int(raw[18]) - r0, int(raw[19]) - r0))
if K == 1 and raw[33] > 0:
    wg = raw[32:32 + 4 * 512].reshape(512, 4).astype(np.int64)
    end = wg[:, 0] - r0
    print("per workgroup (one launch): end tick percentiles 0/10/50/90/100: %s" % np.percentile(end, [0, 10, 50, 90, 100]).astype(int))
    for x in sorted(set(wg[:, 2])):
        sel = wg[:, 2] == x
        print("  XCC %d: %3d workgroups, items %5d, end ticks min %d median %d max %d; first-half WGs items %d, second-half %d"
              % (x, sel.sum(), wg[sel, 1].sum(), end[sel].min(), np.median(end[sel]), end[sel].max(),
                 wg[sel & (np.arange(512) < 256), 1].sum(), wg[sel & (np.arange(512) >= 256), 1].sum()))
if raw[33] > 0 and os.environ.get("GP_STAMP_DUMP"):
    np.save(os.environ["GP_STAMP_DUMP"], raw)
