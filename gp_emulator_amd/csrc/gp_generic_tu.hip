// predict_generic_kernel<T> launcher; build.py compiles it with -DGP_T / -DGP_TNAME.
#include "gp_generic_kernel.hpp"

#define GP_CAT2(a, b) a##b
#define GP_CAT(a, b) GP_CAT2(a, b)

namespace gpk {

hipError_t GP_CAT(launch_generic_, GP_TNAME)(const GenericArgs<GP_T>& a, int grid, hipStream_t stream) {
  const size_t lds = sizeof(GP_T) * (16 * (size_t)(a.N + 1) + 16 * (size_t)a.D);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&predict_generic_kernel<GP_T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((predict_generic_kernel<GP_T>), dim3(grid), dim3(gkThreads), lds, stream, a);
  return hipGetLastError();
}

}  // namespace gpk
