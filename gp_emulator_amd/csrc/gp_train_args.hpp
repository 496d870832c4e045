// Launch geometry, limits and argument block of likelihood_kernel (gp_train_kernel.hpp);
// shared by the C-ABI translation unit, which must not pull in the kernel definition itself.
#pragma once

namespace gpk {

constexpr int tkSide = 32;                  // 32 x 32 thread tile
constexpr int tkThreads = tkSide * tkSide;
constexpr int tkMaxN = 512;                 // LDS: pivot row + column factors + targets
constexpr int tkMaxD = 16;

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
};

}  // namespace gpk
