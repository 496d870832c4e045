// predict_few_kernel<T, D> launchers (every compiled D; NB is a run-time argument); build.py
// compiles this with -DGP_T / -DGP_TNAME.
#include "gp_dispatch.hpp"
#include "gp_predict_few_kernel.hpp"

#define GP_CAT2(a, b) a##b
#define GP_CAT(a, b) GP_CAT2(a, b)

namespace gpk {

hipError_t GP_CAT(launch_few_, GP_TNAME)(int kd, const PredictArgs<GP_T>& a, int nb, int grid, hipStream_t stream) {
  switch (kd) {
#define GP_CASE(d)                                                                                         \
  case d:                                                                                                  \
    hipLaunchKernelGGL((predict_few_kernel<GP_T, d>), dim3(grid), dim3(fkThreads), 0, stream, a, nb);      \
    break;
    GP_FOR_EACH_KERNEL_D(GP_CASE)
#undef GP_CASE
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gpk
