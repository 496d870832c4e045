// predict for FEW test rows: the latency form of predict_kernel.
//
// predict_kernel (gp_predict_kernel.hpp) gives a wave a whole 16-row tile -- phase A over every
// training point, then the tile's 4 NB (NB + 1) / 2 matrix instructions (528 at N = 250) one after
// the other -- because with a million rows there is a tile for every wave of the chip.  A caller
// that asks for one state vector at a time (MultivariateEmulator.predict inside an optimiser,
// gp_emulator/multivariate_gp.py:195-222) launches one tile per emulator and waits ~45 us for that
// single wave.  Here the WORKGROUP owns the tile and its eight waves share the work:
//
//   phase A  (training rows staged in LDS by all threads first) wave w takes k-steps w, w + 8, ...:
//            each lane one (training point, test row) pair per k-step, predict_kernel's arithmetic; the weight tile goes to LDS in the
//            matrix core's B-operand layout ([k-step][lane]), the partial mean / gradient sums of
//            the wave to LDS after its lane-group reduction;
//   phase B  column blocks J are dealt in pairs (J, NB - 1 - J) -- NB + 1 block products per pair,
//            the same for every pair -- and a wave runs k_I^T S'_IJ for I >= J with the A
//            fragments straight from global memory (each fragment is used by one wave of one
//            workgroup: nothing to share through LDS) and the B operands from the LDS tile;
//   finish   wave 0 adds the eight partial results in wave order (deterministic) and stores.
//
// Same packed constants (xa, frags, sd) as predict_kernel; NB is a run-time argument, only the
// row dimension D is compiled in.  Results agree with predict_kernel to rounding (the sums are
// taken in a different order), not bit for bit.
#pragma once
#include "gp_predict_kernel.hpp"

namespace gpk {

constexpr int fkWaves = 8;
constexpr int fkThreads = fkWaves * 64;
constexpr int fkMaxKSteps = 4 * 20;      // 4 GP_MAX_KERNEL_NB

template <typename T, int D>
__global__ __launch_bounds__(fkThreads, 2) void predict_few_kernel(PredictArgs<T> p, int NB) {
  typedef Real<T> R;
  typedef typename R::acc_t acc_t;
  constexpr int DS = row_stride(D);
  constexpr int kVals = D + 2;                       // mu, G_0..G_{D-1}, variance sum

  __shared__ __attribute__((aligned(16))) T s_xa[4 * fkMaxKSteps * row_stride(D)];   // training rows [x'', alpha, h]
  __shared__ T s_k[fkMaxKSteps][64];                 // the weight tile, B-operand layout
  __shared__ T s_part[fkWaves][kTile][kVals];        // per-wave partial sums by test row

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 15, g = lane >> 4;
  const int n_tiles = (int)((p.M + kTile - 1) / kTile);
  const int e = blockIdx.x / n_tiles, tile = blockIdx.x - e * n_tiles;
  const int NKS = 4 * NB;                            // every k-step of the NB blocks (padding rows are zero rows)

  const T* xa = p.xa + e * p.xa_stride;
  const T* frags = p.frags + e * p.frags_stride;
  const T* sdp = p.sd + e * p.sd_stride;
  const long long m = (long long)tile * kTile + ml;
  const long long mc = m < p.M ? m : p.M - 1;

  // training rows -> LDS, all threads, one round of loads in flight
  {
    const int n = 16 * NB * DS;
#pragma unroll 4
    for (int i = tid; i < n; i += fkThreads) s_xa[i] = xa[i];
  }

  // the test row of this lane's column, scaled and centred as predict_kernel does
  T t[D];
  T gm = T(0);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const bool live = d < p.d_actual;
    const int dc = live ? d : p.d_actual - 1;
    const T sc = !live ? T(0) : p.rows_prescaled ? T(1) : sdp[d];
    const T ce = (!live || p.rows_prescaled) ? T(0) : sdp[D + d];
    t[d] = sc * (p.testing[mc * p.d_actual + dc] - ce);
    gm = fma(t[d], t[d], gm);
  }
  gm *= T(-0.5);
  const T poison = gm - gm;                          // NaN for rows holding a NaN or an infinity
  const T b = sdp[2 * D];
  __syncthreads();                                   // training rows visible

  // ---- phase A: this wave's k-steps ------------------------------------------------------------
  T mu = T(0);
  T ga[D];
#pragma unroll
  for (int d = 0; d < D; ++d) ga[d] = T(0);
#pragma unroll 1
  for (int ks = w; ks < NKS; ks += fkWaves) {
    const int i = own_index<T>(ks >> 2, ks & 3, g);
    const T* row = &s_xa[i * DS];
    T x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = row[d];
    const T al = row[D];
    T k;
    if constexpr (R::kExpand) {
      k = row[D + 1] + gm;
#pragma unroll
      for (int d = 0; d < D; ++d) k = fma(x[d], t[d], k);
      k = R::exp_(k);
    } else {
      T r2 = T(0);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        x[d] -= t[d];
        r2 = fma(x[d], x[d], r2);
      }
      k = b * R::exp_(T(-0.5) * r2);
    }
    s_k[ks][lane] = k;
    const T wgt = k * al;
    mu += wgt;
#pragma unroll
    for (int d = 0; d < D; ++d) ga[d] = fma(wgt, x[d], ga[d]);
  }
  mu = xor_reduce_groups(mu);
  xor_reduce_groups_n<T, D>(ga);
  if (g == 0) {
    s_part[w][ml][0] = mu;
#pragma unroll
    for (int d = 0; d < D; ++d) s_part[w][ml][1 + d] = ga[d];
  }
  __syncthreads();                                   // the whole weight tile is in LDS

  // ---- phase B: this wave's column blocks --------------------------------------------------------
  T vacc = T(0);
  auto column_block = [&](const int J) __attribute__((always_inline)) {
    // fragments (I, J, s), I = J .. NB - 1, s = 0 .. 3, lie one after the other in the packed buffer
    const T* f = frags + ((size_t)frag_index(J, J, 0, NB) * 64 + lane);
    const int n = 4 * (NB - J);
    acc_t acc = acc_t{T(0), T(0), T(0), T(0)};
    T a_cur[4], a_nxt[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a_cur[s] = f[(size_t)s * 64];
#pragma unroll 1
    for (int q = 0; q < n; q += 4) {
      const int qn = q + 4 < n ? q + 4 : q;          // (the last round re-reads its own block: no branch)
#pragma unroll
      for (int s = 0; s < 4; ++s) a_nxt[s] = f[(size_t)(qn + s) * 64];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = R::mfma(a_cur[s], s_k[4 * J + q + s][lane], acc);
#pragma unroll
      for (int s = 0; s < 4; ++s) a_cur[s] = a_nxt[s];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) vacc = fma(acc[r], s_k[4 * J + r][lane], vacc);
  };
  for (int pr = w; 2 * pr < NB; pr += fkWaves) {
    column_block(pr);
    if (NB - 1 - pr != pr) column_block(NB - 1 - pr);
  }
  vacc = xor_reduce_groups(vacc);
  if (g == 0) s_part[w][ml][1 + D] = vacc;
  __syncthreads();

  // ---- finish: wave 0 adds the partial results in wave order and stores ---------------------------
  if (w == 0 && m < p.M) {
    // (value by value: fetching all D + 2 sums of all eight waves at once would take ~300 registers)
    auto total = [&](const int v) __attribute__((always_inline)) {
      T s = s_part[0][ml][v];
#pragma unroll
      for (int k = 1; k < fkWaves; ++k) s += s_part[k][ml][v];
      return s;
    };
    const T mean = total(0) + poison;
    T* o_mu = p.mu + e * p.M;
    T* o_var = p.var + e * p.M;
    T* o_der = p.deriv + e * p.M * p.d_actual;
    if (g == 0) o_mu[m] = mean;
    if (g == 1) o_var[m] = b - total(1 + D) + poison;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if ((d & 3) == g && d < p.d_actual) {
        const T gsum = total(1 + d);
        const T gd = sdp[d] * (R::kExpand ? fma(-t[d], mean, gsum) : gsum + poison);
        if (p.deriv_row_major) o_der[m * p.d_actual + d] = gd;
        else o_der[(long long)d * p.M + m] = gd;
      }
    }
  }
}

}  // namespace gpk
