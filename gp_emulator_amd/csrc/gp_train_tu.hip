// likelihood_kernel launcher (fp64 only).
#include "gp_train_kernel.hpp"

namespace gpk {

hipError_t launch_likelihood(const TrainArgs& a, int n_sets, hipStream_t stream) {
  hipLaunchKernelGGL(likelihood_kernel, dim3(n_sets), dim3(tkSide, tkSide), 0, stream, a);
  return hipGetLastError();
}

}  // namespace gpk
