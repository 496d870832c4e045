// Launch geometry, limits and argument block of likelihood_kernel (gp_train_kernel.hpp);
// shared by the C-ABI translation unit, which must not pull in the kernel definition itself.
#pragma once

namespace gpk {

constexpr int tkSide = 32;                  // 32 x 32 thread tile
constexpr int tkThreads = tkSide * tkSide;
constexpr int tkMaxN = 512;                 // LDS: 8 (2 tkB + D) N bytes dynamic + 3 N reals static
constexpr int tkMaxD = 16;
#ifndef GP_TK_U
#define GP_TK_U 4
#endif
constexpr int tkU = GP_TK_U;                // columns per thread per trip of the update loop
constexpr int tkB = 8;                      // pivots per elimination pass

struct TrainArgs {
  const double* theta;     // [E][D + 2]
  const double* inputs;    // [N][D]
  const double* targets;   // [E][N] (targets_stride = N) or [N] shared (targets_stride = 0)
  long long targets_stride;
  double* work;            // [E][N][N]: ends up holding invQ
  double* invQt;           // [E][N]
  double* cost;            // [E]
  double* grad;            // [E][D + 2]
  int N, D;
  int full_inverse;        // 1: `work` must hold the whole invQ (the caller reads it); 0: likelihood_mfma_kernel
                           // writes the lower triangle only (all that invQt and the gradient need)
  unsigned long long* dbg;   // -DGP_TRAIN_STAMPS=1 builds only (tools/train_stamps.py): [8] cycle sums; else unused
};

// dynamic LDS of one workgroup: pivot rows, W, transposed inputs
inline unsigned long likelihood_lds_bytes(int N, int D) {
  return sizeof(double) * (unsigned long)N * (2 * tkB + D);
}

}  // namespace gpk
