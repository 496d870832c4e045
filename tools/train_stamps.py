"""Where a training-objective evaluation spends its cycles (diagnostic build only):

    python -m gp_emulator_amd.build --define GP_TRAIN_STAMPS=1 --lib gp_emulator_amd/libgp_predict_hip_tstamps.so
    GP_PREDICT_LIB=gp_emulator_amd/libgp_predict_hip_tstamps.so python tools/train_stamps.py
(--define GP_TRAIN_STAMPS=2 [--define GP_TRAIN_STAMP_WAVE=4] --only train and GP_TRAIN_STAMPS_FINE=1: the anatomy of a
pass as wave 0 / wave 4 sees it)

Wave 0 of workgroup 0 of likelihood_mfma_kernel sums s_memtime differences per segment
(gp_train_mfma_kernel.hpp, TM_STAMP).  Shares only; never a timing of the real kernel."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_emulator_amd import _lib

N, D, E = 250, 10, int(sys.argv[1]) if len(sys.argv) > 1 else 1
rs = np.random.RandomState(11)
inputs = rs.random_sample((N, D))
targets = np.sin(3 * inputs[:, 0]) + 0.05 * rs.standard_normal(N)
thetas = 0.5 * rs.standard_normal((E, D + 2))
thetas[:, D + 1] -= 4.0
ctx = _lib.Context(0)
d_dbg = ctx.to_device(np.zeros(8, np.uint64))
_lib.check(ctx.lib.gp_ctx_set_debug_buffer(ctx.h, d_dbg))
for _ in range(3):
    ctx.likelihood_batch(thetas, inputs, targets)
seg = ctx.to_host(d_dbg, (8,), np.uint64).astype(np.float64)
names = ["set-up + Q build", "pivot rows -> LDS (+barrier)", "(unused)",
         "panel steps, waves 0-3 (+barrier)", "tile updates (+barrier)", "write-out + invQt", "gradient + sums"]
if os.environ.get("GP_TRAIN_STAMPS_FINE"):    # a --define GP_TRAIN_STAMPS=2 [--define GP_TRAIN_STAMP_WAVE=w] build: a pass's anatomy
    names = ["set-up + Q build", "pivot rows -> LDS", "barrier behind them", "panel steps (+barrier)",
             "update: operand reads", "update: matrix instructions", "update: pivot rows / columns replaced",
             "behind the passes"]
tot = seg[:len(names)].sum()
passes = (N + 7) // 8
print("N=%d D=%d sets=%d: %.0f cycles per evaluation (wave 0 of workgroup 0), %d passes" % (N, D, E, tot, passes))
for n, v in zip(names, seg):
    per = "  (%.0f per pass)" % (v / passes) if 1 <= names.index(n) <= (6 if len(names) == 8 else 4) else ""
    print("  %-34s %9.0f  %5.1f %%%s" % (n, v, 100 * v / tot, per))
