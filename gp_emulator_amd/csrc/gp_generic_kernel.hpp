// General-shape predict kernel: any N_train <= 1024 and n_inputs <= 64.
//
// The fused MFMA kernel (gp_predict_kernel.hpp) is compiled for N_train <= 320 and
// n_inputs <= 16 -- every shape the reference's tests, benchmarks and data use.  Shapes
// beyond that run here: the same arithmetic (gp_emulator/GaussianProcess.py:230-247), the same
// C ABI, still entirely on the GPU, but plain VALU code with run-time loop bounds.  It is a
// correctness path (tens of times slower per point than the MFMA kernel), not a tuned one.
//
// One workgroup (256 threads) per tile of 16 test rows; thread (m = tid >> 4, jl = tid & 15)
// works for test row m and the training points i == jl (mod 16).  The 16 x N kernel-row tile
// lives in LDS (row stride N + 1 reals, so the 16 rows start on different banks); invQ is read
// as given (row-major, no folding) with 16 consecutive columns per 16-lane group.
#pragma once
#include "gp_predict_kernel.hpp"

namespace gpk {

constexpr int gkThreads = 256;
constexpr int gkMaxN = 1024;
constexpr int gkMaxD = 64;

template <typename T>
struct GenericArgs {
  const T* xa;        // [N][ds] training rows: x'' at [0, D), alpha at column acol
  const T* invQ;      // [N][N] row-major, as given by the caller
  const T* sd;        // [2*dk + 1] sqrt(e_d), centre c_d (each dk long), b
  const T* testing;   // [M][D]
  T* mu;
  T* var;
  T* deriv;
  long long M;
  int N, D;
  int ds, acol, dk;   // row stride, alpha column, length of the sd / centre blocks
  int deriv_row_major;
};

template <typename T>
__device__ inline T row16_sum(T v) {   // sum over the 16 lanes that share tid >> 4
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(gkThreads) void predict_generic_kernel(GenericArgs<T> p) {
  typedef Real<T> R;
  extern __shared__ __attribute__((aligned(16))) unsigned char g_raw[];
  T* s_k = reinterpret_cast<T*>(g_raw);            // [16][N + 1]
  const int ns = p.N + 1;
  T* s_t = s_k + 16 * ns;                          // [16][D] scaled, centred test rows

  const int tid = threadIdx.x;
  const int ml = tid >> 4;
  const int jl = tid & 15;
  const T b = p.sd[2 * p.dk];
  const long long n_tiles = (p.M + 15) / 16;

  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long m = tile * 16 + ml;
    const long long mc = m < p.M ? m : p.M - 1;
    __syncthreads();   // previous tile's readers of s_k / s_t are done
    for (int d = jl; d < p.D; d += 16)
      s_t[ml * p.D + d] = p.sd[d] * (p.testing[mc * p.D + d] - p.sd[p.dk + d]);
    __syncthreads();

    // kernel row tile + mean
    const T* t = &s_t[ml * p.D];
    T mu = T(0);
    for (int i = jl; i < p.N; i += 16) {
      const T* row = &p.xa[(long long)i * p.ds];
      T r2 = T(0);
      for (int d = 0; d < p.D; ++d) {
        const T dl = row[d] - t[d];
        r2 = fma(dl, dl, r2);
      }
      const T k = b * R::exp_(T(-0.5) * r2);
      s_k[ml * ns + i] = k;
      mu = fma(k, row[p.acol], mu);
    }
    mu = row16_sum(mu);
    if (jl == 0 && m < p.M) p.mu[m] = mu;

    // gradient, one input dimension at a time
    for (int d = 0; d < p.D; ++d) {
      T gsum = T(0);
      for (int i = jl; i < p.N; i += 16) {
        const T* row = &p.xa[(long long)i * p.ds];
        gsum = fma(s_k[ml * ns + i] * row[p.acol], row[d] - t[d], gsum);
      }
      gsum = row16_sum(gsum) * p.sd[d];
      if (jl == 0 && m < p.M) {
        if (p.deriv_row_major) p.deriv[m * p.D + d] = gsum;
        else p.deriv[(long long)d * p.M + m] = gsum;
      }
    }
    __syncthreads();   // the whole kernel-row tile is in LDS

    // variance: k^T invQ k with invQ as given (16 consecutive columns per lane group)
    const T* kr = &s_k[ml * ns];
    T acc = T(0);
    for (int j = jl; j < p.N; j += 16) {
      T s = T(0);
      for (int i = 0; i < p.N; ++i) s = fma(p.invQ[(long long)i * p.N + j], kr[i], s);
      acc = fma(s, kr[j], acc);
    }
    acc = row16_sum(acc);
    if (jl == 0 && m < p.M) p.var[m] = b - acc;
  }
}

}  // namespace gpk
