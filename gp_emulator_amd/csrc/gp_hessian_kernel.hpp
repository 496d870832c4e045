// Hessian of the GP mean on gfx950: hess[m][d][d2] for every test row m.
//
// The reference has only a numpy version, gp_emulator/GaussianProcess.py:345-366, which
// makes D^2 passes over an (N_test, N_train) array.  Here it is one fused pass: with
// w_i = k_i alpha_i and delta_i = x''_i - t'' (sqrt(e)-scaled, centred coordinates)
//
//   hess[d][d2] = sd_d sd_d2 * sum_i w_i delta_id delta_id2  -  [d == d2] e_d * sum_i w_i
//
// i.e. a symmetric rank-N update per test row, done here on the VALU with D(D+1)/2 fmas per
// (training point, test row).  This is the kernel for n_inputs <= 5; from kernel D = 8 up the
// same quantity is computed on the matrix core by gp_hessian_mfma_kernel.hpp (the pair products
// x''_id x''_id2 do not depend on the test row, which turns the sum into a matrix product), which
// is 1.2-1.5x faster at D = 8, 12 and 16 (even at D = 10, 11 in fp64, where 4 x 4 blocks pad 55 / 66
// products to 96).  GP_HESS_VALU=1 in the environment forces this kernel for every D.
//
// Lane layout as in the predict kernel: lane l works for test row (l & 15) and the quarter
// (l >> 4) of the training points; the D(D+1)/2 partial sums live in registers (one wave
// per SIMD: up to 136 fp64 accumulators), are summed over the four lane groups at the end,
// and each lane group writes a quarter of the rows of the (D, D) block.  The loop over
// training points is a real loop (nothing is indexed by it), so the code stays small.
#pragma once
#include "gp_predict_kernel.hpp"

#ifndef GP_HESS_ABLATE
#define GP_HESS_ABLATE 0
#endif

namespace gpk {

template <typename T>
struct HessianArgs {
  const T* xa;        // [16*nb][row_stride(D)] training rows [x'', alpha, h]
  const T* sd;        // [2*D + 1] sqrt(e_d), the centre c_d, b = e[D]
  const T* testing;   // [M][d_actual]
  T* hess;            // [M][d_actual][d_actual]
  long long M;
  int d_actual;
  int nb;             // 16-row blocks of training points
};

// Hessian kernel geometry: 4 waves (one per SIMD) so that the D = 16 instance, which needs
// ~380 registers, still fits (one wave per SIMD).
constexpr int hkWaves = 4;   // two workgroups per CU = two waves per SIMD
constexpr int hkThreads = hkWaves * 64;
constexpr int hkRowsPerWG = hkWaves * kTile;

// Rows [lo, hi) of the upper triangle handled in one pass: the pass owns the pairs
// (d, d2 >= d) with lo <= d < hi.
__host__ __device__ constexpr int pairs_in_rows(int D, int lo, int hi) {
  int n = 0;
  for (int d = lo; d < hi; ++d) n += D - d;
  return n;
}
// Split row so that both passes hold about half of the D(D+1)/2 accumulators.
__host__ __device__ constexpr int split_row(int D) {
  int best = 0, bestdiff = 1 << 30;
  for (int r = 0; r <= D; ++r) {
    int a = pairs_in_rows(D, 0, r), b = pairs_in_rows(D, r, D);
    int diff = a > b ? a - b : b - a;
    if (diff < bestdiff) { bestdiff = diff; best = r; }
  }
  return best;
}

// One pass over the training points accumulating rows [LO, HI) of the Hessian's upper
// triangle, then the reduction over lane groups and the stores (with mirror elements).
template <typename T, int D, int LO, int HI>
__device__ __forceinline__ void hessian_pass(const T* s_xa, const T* s_sd, const T (&t)[D], T b,
                                             int nb, int g, bool row_ok, int d_actual, T* out) {
  typedef Real<T> R;
  constexpr int DS = row_stride(D);
  constexpr int NP_ = pairs_in_rows(D, LO, HI);
  T acc[NP_];
#pragma unroll
  for (int q = 0; q < NP_; ++q) acc[q] = T(0);
  T mu = T(0);
  // One training point per iteration, NOT unrolled (unrolling lets the scheduler keep several
  // points' delta vectors live on top of the accumulators and spill).  The NEXT point's row is
  // read from LDS while this one's products issue, and the squared distance is summed in four
  // short chains: with two waves per SIMD the loop is otherwise latency-bound (LDS read -> 16
  // dependent fmas -> exp before the first of the independent products can start).
  T nxt[D + 1];
  {
    const T* row = &s_xa[R::own_sub(0, g) * DS];
#pragma unroll
    for (int d = 0; d <= D; ++d) nxt[d] = row[d];
  }
#pragma unroll 1
  for (int qq = 0; qq < 4 * nb; ++qq) {
    T dl[D];
    T r2p[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int d = 0; d < D; ++d) {
      dl[d] = nxt[d] - t[d];
      r2p[d & 3] = fma(dl[d], dl[d], r2p[d & 3]);
    }
    const T alpha = nxt[D];
    {   // prefetch (the row after the last is row 0 again: harmless, in range)
      const int qn = (qq + 1 < 4 * nb) ? qq + 1 : 0;
      const T* row = &s_xa[(16 * (qn >> 2) + R::own_sub(qn & 3, g)) * DS];
#pragma unroll
      for (int d = 0; d <= D; ++d) nxt[d] = row[d];
    }
    const T r2 = (r2p[0] + r2p[1]) + (r2p[2] + r2p[3]);
    const T w = b * R::exp_(T(-0.5) * r2) * alpha;
    mu += w;
    int q = 0;
#pragma unroll
    for (int d = LO; d < HI; ++d) {
      const T wd = w * dl[d];
#pragma unroll
      for (int d2 = d; d2 < D; ++d2) {
        acc[q] = fma(wd, dl[d2], acc[q]);
        ++q;
      }
    }
  }
#if GP_HESS_ABLATE == 2   // diagnostic: no reductions, no stores (one guarded store keeps the sums alive)
  T keep = mu;
#pragma unroll
  for (int q = 0; q < NP_; ++q) keep += acc[q];
  if (keep == T(-12345.678)) out[0] = keep;
  return;
#endif
  mu = xor_reduce_groups(mu);
  int q = 0;
#pragma unroll
  for (int d = LO; d < HI; ++d) {
#pragma unroll
    for (int d2 = d; d2 < D; ++d2) {
      T v = xor_reduce_groups(acc[q]) * (s_sd[d] * s_sd[d2]);
      if (d2 == d) v = fma(-(s_sd[d] * s_sd[d]), mu, v);
      ++q;
#if GP_HESS_ABLATE == 1   // diagnostic: reductions but no stores
      if (v == T(-12345.678)) out[0] = v;
#else
      if (row_ok && d < d_actual && d2 < d_actual) {
        // element (d, d2) by lane group d & 3, its mirror (d2, d) by group d2 & 3
        if ((d & 3) == g) out[d * d_actual + d2] = v;
        if (d2 != d && (d2 & 3) == g) out[d2 * d_actual + d] = v;
      }
#endif
    }
  }
}

template <typename T, int D>
__global__ __launch_bounds__(hkThreads, 2) void hessian_kernel(HessianArgs<T> p) {
  typedef Real<T> R;
  constexpr int DS = row_stride(D);
  constexpr int NPAIR = D * (D + 1) / 2;
  // two passes when the accumulators alone would exceed ~half the register file
  constexpr int kSplit = (NPAIR * (int)sizeof(T) / 4 > 144) ? split_row(D) : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  T* s_xa = reinterpret_cast<T*>(s_raw);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 15;
  const int g = lane >> 4;
  const int np = 16 * p.nb;

  for (int i = tid; i < np * DS; i += hkThreads) s_xa[i] = p.xa[i];
  // sqrt(e_d) and the centre stay in LDS (broadcast reads), not in registers: the
  // D(D+1)/2 accumulators need the register file
  T* s_sd = s_xa + np * DS;
  if (tid < 2 * D) s_sd[tid] = ((tid % D) < p.d_actual) ? p.sd[tid] : T(0);
  const T b = p.sd[2 * D];
  __syncthreads();

  const long long n_groups = (p.M + hkRowsPerWG - 1) / hkRowsPerWG;
  for (long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long long m = grp * hkRowsPerWG + wave * kTile + ml;
    const long long mc = m < p.M ? m : p.M - 1;
    T t[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
      t[d] = (d < p.d_actual) ? s_sd[d] * (p.testing[mc * p.d_actual + d] - s_sd[D + d]) : T(0);
    T* out = p.hess + mc * (long long)p.d_actual * p.d_actual;
    const bool row_ok = m < p.M;
    // D(D+1)/2 fp64 accumulators do not fit the VALU-addressable registers beyond D ~ 11
    // (the compiler parks the excess in AGPRs and pays two moves per fma); above that the
    // triangle is done in two passes of ~half the rows each, recomputing the kernel row.
    if constexpr (kSplit > 0) {
      hessian_pass<T, D, 0, kSplit>(s_xa, s_sd, t, b, p.nb, g, row_ok, p.d_actual, out);
      hessian_pass<T, D, kSplit, D>(s_xa, s_sd, t, b, p.nb, g, row_ok, p.d_actual, out);
    } else {
      hessian_pass<T, D, 0, D>(s_xa, s_sd, t, b, p.nb, g, row_ok, p.d_actual, out);
    }
  }
}

}  // namespace gpk
