// reconstruct_kernel<T, PT, THREADS, NV> launchers: PT in {4, 8, 12, 16}, two geometries;
// both dtypes in one unit.
#include "gp_reconstruct_kernel.hpp"

namespace gpk {

template <typename T, int TH, int NV>
static hipError_t launch_geo(const ReconArgs<T>& a, int grid, hipStream_t stream) {
  if (a.P <= 4) hipLaunchKernelGGL((reconstruct_kernel<T, 4, TH, NV>), dim3(grid), dim3(TH), 0, stream, a);
  else if (a.P <= 8) hipLaunchKernelGGL((reconstruct_kernel<T, 8, TH, NV>), dim3(grid), dim3(TH), 0, stream, a);
  else if (a.P <= 12) hipLaunchKernelGGL((reconstruct_kernel<T, 12, TH, NV>), dim3(grid), dim3(TH), 0, stream, a);
  else if (a.P <= 16) hipLaunchKernelGGL((reconstruct_kernel<T, 16, TH, NV>), dim3(grid), dim3(TH), 0, stream, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// wide = 1: the (512, 3) geometry; returns the bands one workgroup covers through *bw
template <typename T>
static hipError_t launch_recon(const ReconArgs<T>& a, int wide, int cus, hipStream_t stream) {
  const long long bw = wide ? recon_bands_per_wg<T, 512, 3>() : recon_bands_per_wg<T, 256, 2>();
  const long long chunks = (a.B + bw - 1) / bw;
  const long long items = (a.R + rkRows - 1) / rkRows * chunks;
  // memory-bound: a few blocks per CU, grid-stride; every block gets the same number of items
  // (+-1) so that no block is left with a whole extra round at the end
  const long long cap = (long long)cus * (wide ? 4 : 8);
  const long long rounds = (items + cap - 1) / cap;
  const int grid = (int)((items + rounds - 1) / rounds);
  return wide ? launch_geo<T, 512, 3>(a, grid, stream) : launch_geo<T, 256, 2>(a, grid, stream);
}

hipError_t launch_reconstruct_f32(const ReconArgs<float>& a, int wide, int cus, hipStream_t s) { return launch_recon<float>(a, wide, cus, s); }
hipError_t launch_reconstruct_f64(const ReconArgs<double>& a, int wide, int cus, hipStream_t s) { return launch_recon<double>(a, wide, cus, s); }

}  // namespace gpk
