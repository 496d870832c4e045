// One translation unit per compute dtype: hessian_kernel<T, D> for every supported kernel D.
// build.py compiles it with -DGP_T=<float|double> -DGP_TNAME=<f32|f64>.
#include "gp_hessian_kernel.hpp"
#include "gp_dispatch.hpp"

#define GP_CAT2(a, b) a##b
#define GP_CAT(a, b) GP_CAT2(a, b)

namespace gpk {

template <int D>
static hipError_t launch_one(const HessianArgs<GP_T>& a, int grid, size_t lds, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hessian_kernel<GP_T, D>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((hessian_kernel<GP_T, D>), dim3(grid), dim3(hkThreads), lds, stream, a);
  return hipGetLastError();
}

hipError_t GP_CAT(launch_hessian_, GP_TNAME)(int kernel_d, const HessianArgs<GP_T>& a, int grid,
                                             hipStream_t stream) {
  const size_t lds = sizeof(GP_T) * (16 * (size_t)a.nb * row_stride(kernel_d) + 2 * kernel_d);
  switch (kernel_d) {
#define GP_CASE(d) case d: return launch_one<d>(a, grid, lds, stream);
    GP_FOR_EACH_KERNEL_D(GP_CASE)
#undef GP_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace gpk
