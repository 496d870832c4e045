// likelihood kernels' launcher (fp64 only).
#include <cstdlib>

#include "gp_train_kernel.hpp"
#include "gp_train_mfma_kernel.hpp"

namespace gpk {

template <typename K>
static hipError_t allow_lds(K kernel, size_t lds) {
  if (lds <= 48 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
}

hipError_t launch_likelihood(const TrainArgs& a, int n_sets, hipStream_t stream) {
  const size_t lds = likelihood_lds_bytes(a.N, a.D);
  const size_t lds_grad = sizeof(double) * (size_t)a.N * a.D;
  const dim3 grid(n_sets), block(tkSide, tkSide);
  hipError_t e = hipSuccess;
  // n_train <= 256: the register-resident matrix-core elimination (gp_train_mfma_kernel.hpp);
  // larger sets, or GP_TRAIN_GENERIC=1 (A/B switch): the workspace kernel
  static const bool generic = [] { const char* ev = getenv("GP_TRAIN_GENERIC"); return ev && atoi(ev) != 0; }();
  if (a.N <= tmNP && !generic) {
#define GP_LIK(DM) hipLaunchKernelGGL(likelihood_mfma_kernel<DM>, grid, dim3(tmThreads), 0, stream, a)
    if (a.D <= 4) GP_LIK(4);
    else if (a.D <= 8) GP_LIK(8);
    else if (a.D <= 12) GP_LIK(12);
    else GP_LIK(tkMaxD);
#undef GP_LIK
    return hipGetLastError();            // (the gradient is part of that kernel)
  } else {
    e = allow_lds(likelihood_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(likelihood_kernel, grid, block, lds, stream, a);
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
#define GP_GRAD(DM)                                                                  \
  do {                                                                               \
    if ((e = allow_lds(likelihood_grad_kernel<DM>, lds_grad)) != hipSuccess) return e; \
    hipLaunchKernelGGL(likelihood_grad_kernel<DM>, grid, block, lds_grad, stream, a);  \
  } while (0)
  if (a.D <= 4) GP_GRAD(4);
  else if (a.D <= 8) GP_GRAD(8);
  else if (a.D <= 12) GP_GRAD(12);
  else GP_GRAD(tkMaxD);
#undef GP_GRAD
  return hipGetLastError();
}

}  // namespace gpk
