// One translation unit per (compute dtype, NK): instantiates predict_kernel<T, D, NK> for
// every supported kernel D and exports one launcher.  Compiled several times by build.py
// with -DGP_T=<float|double> -DGP_TNAME=<f32|f64> -DGP_NK=<groups of 4 training points>.
#include "gp_predict_kernel.hpp"
#include "gp_dispatch.hpp"

#define GP_CAT2(a, b, c) a##b##_##c
#define GP_CAT(a, b, c) GP_CAT2(a, b, c)

namespace gpk {

template <int D>
static hipError_t launch_one(const PredictArgs<GP_T>& a, int grid, hipStream_t stream) {
  hipLaunchKernelGGL((predict_kernel<GP_T, D, GP_NK>), dim3(grid), dim3(Geo<GP_T>::kThreads), 0, stream, a);
  return hipGetLastError();
}

hipError_t GP_CAT(launch_predict_, GP_TNAME, GP_NK)(int kernel_d, const PredictArgs<GP_T>& a,
                                                   int grid, hipStream_t stream) {
  switch (kernel_d) {
#define GP_CASE(d) case d: return launch_one<d>(a, grid, stream);
    GP_FOR_EACH_KERNEL_D(GP_CASE)
#undef GP_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace gpk
