// One translation unit per (compute dtype, NB): hessian_mfma_kernel<T, D, NB> for the kernel
// dimensions where the matrix-core form beats (or ties with) the VALU one (D >= 8;
// profiles/r01_hessian_kernels.txt).  Compiled by build.py with
// -DGP_T=<float|double> -DGP_TNAME=<f32|f64> -DGP_NB=<blocks of 16 training points>.
#include <stdlib.h>
#include "gp_hessian_mfma_kernel.hpp"
#include "gp_hessian_win_kernel.hpp"

#define GP_CAT2(a, b, c) a##b##_##c
#define GP_CAT(a, b, c) GP_CAT2(a, b, c)

namespace gpk {

// GP_HESS_DIRECT=1 (read once): the direct stores of round 2 for every call, the A/B reference of the
// whole-line finish
static bool hess_win_direct_stores() {
  static const bool v = [] { const char* ev = getenv("GP_HESS_DIRECT"); return ev && atoi(ev) != 0; }();
  return v;
}

// grid: the persistent grid's size for a call that fills the chip (workgroups per CU x CUs); a launch takes
// min(grid, its 64-row groups)
template <int D>
static hipError_t launch_one(const HessMfmaArgs<GP_T>& a, int grid, hipStream_t stream) {
  if constexpr (hess_win<GP_T>(D, GP_NB)) {
    if (a.use_win) {      // the windowed form (gp_hessian_win_kernel.hpp): k-step-major fragments, 4-wave workgroups
      constexpr int kRows = WGeo::kRowsPerWG;
      auto groups_of = [&](long long m) { const long long g_ = (m + kRows - 1) / kRows; return (int)(g_ < grid ? g_ : grid); };
      HessMfmaArgs<GP_T> rest = a;
      if constexpr (win_lds_out<GP_T>(D)) {
        // whole-line stores through LDS when the caller's rows are exactly D long and the matrix is aligned: the
        // whole 64-row groups of the call; what is left (< 64 rows) goes to the direct-store instance below
        const long long m_main = a.M / kRows * kRows;
        if (m_main > 0 && a.d_actual == D && (((unsigned long long)a.hess | (unsigned long long)a.testing) & 15) == 0 &&
            !hess_win_direct_stores()) {
          HessMfmaArgs<GP_T> b = a;
          b.M = m_main;
          bool done = false;
          if constexpr (hess_win_short_last<GP_T>(GP_NB)) {
            if (a.n_ksteps == 4 * GP_NB - 1) {      // the last k-step holds nothing but padding: not issued
              hipLaunchKernelGGL((hessian_win_kernel<GP_T, D, GP_NB, true, 3>), dim3(groups_of(m_main)), dim3(WGeo::kThreads), 0, stream, b);
              done = true;
            }
          }
          if (!done)
            hipLaunchKernelGGL((hessian_win_kernel<GP_T, D, GP_NB, true>), dim3(groups_of(m_main)), dim3(WGeo::kThreads), 0, stream, b);
          hipError_t e = hipGetLastError();
          if (e != hipSuccess || m_main == a.M) return e;
          rest.M = a.M - m_main;
          rest.testing = a.testing + m_main * D;
          rest.hess = a.hess + m_main * D * D;
          rest.tickets = a.tickets2;
        }
      }
      hipLaunchKernelGGL((hessian_win_kernel<GP_T, D, GP_NB, false>), dim3(groups_of(rest.M)), dim3(WGeo::kThreads), 0, stream, rest);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((hessian_mfma_kernel<GP_T, D, GP_NB>), dim3(grid), dim3(HGeo<GP_T, D, GP_NB>::kThreads), 0, stream, a);
  return hipGetLastError();
}

hipError_t GP_CAT(launch_hessm_, GP_TNAME, GP_NB)(int kernel_d, const HessMfmaArgs<GP_T>& a,
                                                 int grid, hipStream_t stream) {
  switch (kernel_d) {
    case 8: return launch_one<8>(a, grid, stream);
    case 10: return launch_one<10>(a, grid, stream);
    case 11: return launch_one<11>(a, grid, stream);
    case 12: return launch_one<12>(a, grid, stream);
    case 16: return launch_one<16>(a, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace gpk
