// Hessian of the GP mean on the matrix core.
//
// The reference (gp_emulator/GaussianProcess.py:345-366) forms, for every test row t,
//   hess[d][d2] = sum_i w_i (e_d (x_id - t_d) e_d2 (x_id2 - t_d2) - [d == d2] e_d),   w_i = k_i alpha_i.
// In the kernel's scaled, centred coordinates x'' = sqrt(e)(x - c) the sum expands to
//   sum_i w_i (x''_id - t''_d)(x''_id2 - t''_d2) = S2[d][d2] - G_d t''_d2 - t''_d G_d2 + s t''_d t''_d2,
//   S2[d][d2] = sum_i w_i x''_id x''_id2,   G_d = sum_i w_i x''_id,   s = sum_i w_i,
// and the products P[i][(d, d2)] = x''_id x''_id2 do NOT depend on the test row: S2 = W P is a
// matrix product of the (test rows x N) weight tile with a constant N x D(D+1)/2 matrix -- the
// same shape as the variance contraction of predict_kernel, and it runs the same way:
//
//   phase A (VALU)  the weight tile w_i = k_i alpha_i of the wave's 16 test rows, in registers in
//                   the matrix core's B-operand layout, with s and G_d (what predict_kernel
//                   computes for the mean and the gradient);
//   phase B (MFMA)  for each block of 16 (d, d2) pairs, 4 NB MFMAs  acc += P-fragment x w  with
//                   the packed P fragments streamed L2 -> LDS by LDS-DMA, double-buffered and
//                   shared by the workgroup's waves; the 4 accumulator registers of a lane then
//                   hold S2 of 4 pairs for the lane's own test row, and the lane finishes and
//                   stores those elements (and their mirror images).
//
// Against the VALU kernel (gp_hessian_kernel.hpp: the same sum as D(D+1)/2 fmas per training
// point and test row, two passes, 136 cross-lane reductions per row at D = 16) this needs no
// reductions for the pair sums at all and moves the bulk of the arithmetic to the matrix pipe,
// which the compiler-scheduled VALU loop only fills to ~50 %.  The pairs are taken in 4 x 4 blocks
// of the matrix (160 products instead of 136 at D = 16, 96 instead of 66 at D = 11).
//
// Two geometries.  The standard one is predict_kernel's (fp64: 8 waves, two per SIMD, 256 registers
// each).  The fp64 weight tile alone is 8 NB registers (152 at N = 300), and with t'', G and a
// training row beside it phase A of the larger instances does not fit 256: they spilled up to 141
// registers and had to give up the 16-byte stores.  Those instances (hess_wide) run ONE wave per
// SIMD with the whole 512-register file (4 waves of 16 rows per workgroup): the weights sit in the
// accumulation half, which the matrix instructions read as B operands directly, nothing spills,
// and every path stores 16 bytes.  A single wave's dependent chain of fp64 matrix instructions
// reaches 79 % of the pipe and two interleaved chains 85 % (tools/mfma_f64_probe.hip), so the
// wide kernel works on two pair blocks at a time, their fragments interleaved in the packed
// buffer (hess_frag_index, paired order).
//
// Accuracy: the expansion cancels like predict_kernel's exp(h_i + g + x''.t'') does -- by about
// (|x''| + |t''|)^2 / |x'' - t''|^2, small because the coordinates are centred on the training
// mean; fp64 stays at 1e-15 on the benchmark sets (tests).
#pragma once
#include "gp_predict_kernel.hpp"

#ifndef GP_HESS_GROUP
#define GP_HESS_GROUP 4
#endif
#ifndef GP_HESSM_ABLATE
#define GP_HESSM_ABLATE 0
#endif

namespace gpk {

// The D x D matrix is cut into 4 x 4 blocks (bi, bj); the blocks with bi <= bj are the 16-wide
// column blocks of P.  Within a block, accumulator register r of lane group g (MFMA output row
// own_sub(r, g)) is element (d, d2) = (4 bi + r, 4 bj + g): the first index is known at compile
// time wherever the accumulator is used, the second is the lane group's own -- so the epilogue
// needs no lookup tables (t''_d2, G_d2, sqrt(e_d2) are three small per-lane arrays indexed by bj).
// Diagonal blocks carry both (d, d2) and (d2, d); D is padded to a multiple of 4 (zero products).
__host__ __device__ constexpr int hess_nb4(int D) { return (D + 3) / 4; }
__host__ __device__ constexpr int hess_blocks(int D) { return hess_nb4(D) * (hess_nb4(D) + 1) / 2; }
__host__ __device__ constexpr int hess_block_index(int bi, int bj) { return bj * (bj + 1) / 2 + bi; }
__host__ __device__ constexpr int hess_block_bj(int c) {
  int bj = 0;
  while ((bj + 1) * (bj + 2) / 2 <= c) ++bj;
  return bj;
}
__host__ __device__ constexpr int hess_block_bi(int c) { return c - hess_block_bj(c) * (hess_block_bj(c) + 1) / 2; }
// Instances that run one wave per SIMD with 512 registers (see the header comment).
template <typename T> __host__ __device__ constexpr bool hess_wide(int D, int NB) {
  return sizeof(T) == 8 && (NB >= 16 || (NB >= 12 && D >= 16));
}
// fragment (block c, training block I, k-step s) in consumption order.  Plain order: block after
// block.  Paired order (wide kernels): blocks 2p and 2p + 1 alternate fragment by fragment, so
// two independent accumulator chains are in flight; an odd last block comes alone.
__host__ __device__ constexpr int hess_frag_index(int c, int I, int s, int NB, bool paired = false, int nblk = 0) {
  if (!paired || (c == nblk - 1 && (nblk & 1))) return (c * NB + I) * 4 + s;
  return (c / 2) * 8 * NB + (I * 4 + s) * 2 + (c & 1);
}
struct HessFragId { int c, I, s; };
__host__ __device__ constexpr HessFragId hess_frag_at(int n, int NB, bool paired, int nblk) {
  if (!paired || n >= (nblk / 2) * 8 * NB) return HessFragId{n / (4 * NB), (n / 4) % NB, n % 4};
  const int p = n / (8 * NB), r = n % (8 * NB);
  return HessFragId{2 * p + (r & 1), (r / 2) / 4, (r / 2) % 4};
}
__host__ __device__ constexpr int hess_frag_count(int D, int NB) { return hess_blocks(D) * NB * 4; }
__host__ __device__ constexpr int hess_frag_count_padded(int D, int NB, int chunk) {
  return (hess_frag_count(D, NB) + chunk - 1) / chunk * chunk;
}
// (r, g) with own_sub(r, g) == q: which accumulator register / lane group MFMA output row q is
template <typename T> __host__ __device__ constexpr int hess_row_r(int q) {
  for (int r = 0; r < 4; ++r)
    for (int g = 0; g < 4; ++g)
      if (Real<T>::own_sub(r, g) == q) return r;
  return 0;
}
template <typename T> __host__ __device__ constexpr int hess_row_g(int q) {
  for (int r = 0; r < 4; ++r)
    for (int g = 0; g < 4; ++g)
      if (Real<T>::own_sub(r, g) == q) return g;
  return 0;
}

// 4 x 4 transpose between accumulator register r and lane group g: v[r] of group g <-> v[g] of
// group r.  v_permlane32_swap exchanges the upper half of one register with the lower half of
// another (a 2 x 2 transpose on (r bit 1, g bit 1)), v_permlane16_swap the odd 16-lane rows of one
// with the even rows of the other ((r bit 0, g bit 0)).
__device__ __forceinline__ void swap_halves2(unsigned& a, unsigned& b, bool rows16) {
  if (rows16) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
  }
}
__device__ __forceinline__ void swap_pair(float& a, float& b, bool rows16) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  swap_halves2(ua, ub, rows16);
  a = __uint_as_float(ua);
  b = __uint_as_float(ub);
}
__device__ __forceinline__ void swap_pair(double& a, double& b, bool rows16) {
  unsigned long long xa = (unsigned long long)__double_as_longlong(a), xb = (unsigned long long)__double_as_longlong(b);
  unsigned alo = (unsigned)xa, ahi = (unsigned)(xa >> 32), blo = (unsigned)xb, bhi = (unsigned)(xb >> 32);
  swap_halves2(alo, blo, rows16);
  swap_halves2(ahi, bhi, rows16);
  a = __longlong_as_double((long long)(((unsigned long long)ahi << 32) | alo));
  b = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
}
template <typename T>
__device__ __forceinline__ void transpose_groups4(T (&v)[4]) {
  swap_pair(v[0], v[2], false);
  swap_pair(v[1], v[3], false);
  swap_pair(v[0], v[1], true);
  swap_pair(v[2], v[3], true);
}
// four consecutive elements of one output row as 16-byte stores; `left` = elements of the row
// still inside the matrix from this position on (a multiple of the vector width, possibly <= 0)
template <typename T>
__device__ __forceinline__ void store_row4(T* dst, const T (&v)[4], int left) {
  if constexpr (sizeof(T) == 8) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    if (left > 0) *reinterpret_cast<d2_t*>(dst) = d2_t{v[0], v[1]};
    if (left > 2) *reinterpret_cast<d2_t*>(dst + 2) = d2_t{v[2], v[3]};
  } else {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    if (left > 0) *reinterpret_cast<f4_t*>(dst) = f4_t{v[0], v[1], v[2], v[3]};
  }
}

template <typename T>
struct HessMfmaArgs {
  const T* xa;        // [16*NB][row_stride(D)] training rows [x'', alpha, h]
  const T* pfrags;    // [hess_frag_count_padded][64] products x''_id x''_id2, 4 x 4 blocks, fragment order
  const T* sd;        // [2*D + 1] sqrt(e_d), the centre c_d, b = e[D]
  const T* testing;   // [M][d_actual]
  T* hess;            // [M][d_actual][d_actual]
  long long M;
  int d_actual;
  int use_win;        // hess_wide instances: 1 = hessian_win_kernel (pfrags packed k-step-major), 0 = the wide geometry
  unsigned long long* dbg;   // GP_STAMPS builds only (hessian_win_kernel): [8] segment cycle sums; else unused
  int n_ksteps;              // ceil(n_train / 4): k-steps that hold training points
  unsigned* tickets2;        // host side only: the counter of the call's second launch (rows beyond the last whole 64-row group)
  unsigned* tickets;         // hessian_win_kernel: the launch's item counter (0 on entry and on exit), or null: items dealt round-robin
};

// Geometry of hessian_mfma_kernel<T, D, NB> (see the header comment).
template <typename T, int D, int NB> struct HGeo {
  static constexpr bool kWide = hess_wide<T>(D, NB);
  static constexpr int kWaves = kWide ? 4 : Geo<T>::kWaves;
  static constexpr int kThreads = kWaves * 64;
  static constexpr int kWavesPerSimd = kWide ? 1 : Geo<T>::kWavesPerSimd;
  static constexpr int kRowsPerWG = kWaves * kTile;
  // A-operand fragments per LDS chunk.  Wide: one pair block's worth (4 NB), so that a chunk
  // boundary -- whose s_waitcnt vmcnt(0) also waits for the wave's outstanding global stores --
  // always comes a whole block after the previous block's stores were issued.
  static constexpr int kChunk = kWide ? 4 * NB : Geo<T>::kChunk;
  static constexpr int kGroup = kWide ? GP_HESS_GROUP : 1;     // training points in flight in phase A
  static constexpr int kAccs = kWide ? 4 : 2;      // accumulators (blocks alternate; wide: pairs)
};

template <typename T, int D, int NB>
__global__ __launch_bounds__((HGeo<T, D, NB>::kThreads), (HGeo<T, D, NB>::kWavesPerSimd))
void hessian_mfma_kernel(HessMfmaArgs<T> p) {
  typedef Real<T> R;
  typedef typename R::acc_t acc_t;
  typedef HGeo<T, D, NB> G;
  constexpr bool kWide = G::kWide;
  constexpr int kThreads = G::kThreads;
  constexpr int kWaves = G::kWaves;
  constexpr int kRowsPerWG = G::kRowsPerWG;
  constexpr int NP = 16 * NB;
  constexpr int DS = row_stride(D);
  constexpr int NB4 = hess_nb4(D);
  constexpr int NBLK = hess_blocks(D);
  constexpr int NF = hess_frag_count(D, NB);
  constexpr int kChunk = G::kChunk;
  constexpr int NCH = (NF + kChunk - 1) / kChunk;

  __shared__ __attribute__((aligned(16))) T s_xa[NP * DS];
  __shared__ __attribute__((aligned(16))) T s_fr[2][kChunk * 64];
  __shared__ T s_sd[2 * D + 1];
  // per test row of the wave: t''_d, then G_d.  Row stride 2 D + 1: the epilogue reads one d for
  // the 16 rows at once, and a stride of 2 D = 32 doubles would put them all in the same bank
  __shared__ T s_tg[kWaves][kTile][2 * D + 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 15;
  const int g = lane >> 4;

  for (int i = tid; i < NP * DS; i += kThreads) s_xa[i] = p.xa[i];
  if (tid < 2 * D + 1) s_sd[tid] = (tid == 2 * D || (tid % D) < p.d_actual) ? p.sd[tid] : T(0);
  __syncthreads();
  const T b = s_sd[2 * D];
  T sdq[NB4];                   // sqrt(e) of this lane group's column in every block: d2 = 4 bj + g
#pragma unroll
  for (int j = 0; j < NB4; ++j) sdq[j] = (4 * j + g < D) ? s_sd[4 * j + g] : T(0);

  const long long n_groups = (p.M + kRowsPerWG - 1) / kRowsPerWG;
  // the lane's raw test row: loaded one item ahead (wide kernels: a lone wave per SIMD has
  // nothing else to hide the HBM latency of these loads behind; they land during phase A)
  T rraw[D];
  auto load_row = [&](long long grp_) {
    const long long m_ = grp_ * kRowsPerWG + wave * kTile + ml;
    const long long mc_ = m_ < p.M ? m_ : p.M - 1;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int dc = d < p.d_actual ? d : p.d_actual - 1;     // padded dims: sd = centre = 0
      rraw[d] = p.testing[mc_ * p.d_actual + dc];
    }
  };
  if constexpr (kWide)
    if ((long long)blockIdx.x < n_groups) load_row(blockIdx.x);
  for (long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long long m = grp * kRowsPerWG + wave * kTile + ml;
    const long long mc = m < p.M ? m : p.M - 1;
    __syncthreads();            // previous item's readers of s_fr[0] are done
    stage_chunk<T, kWaves, kChunk>(p.pfrags, &s_fr[0][0], wave, lane);

    // ---------------- phase A: weight tile, s, G ----------------------------------------
    T t[D];
    T gm = T(0);
    if constexpr (!kWide) load_row(grp);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      t[d] = s_sd[d] * (rraw[d] - s_sd[D + d]);
      gm = fma(t[d], t[d], gm);
    }
    if constexpr (kWide) {
      asm volatile("" :: "v"(gm));       // (the loads below stay below the uses above)
      load_row(grp + gridDim.x < n_groups ? grp + gridDim.x : grp);
    }
    gm *= T(-0.5);
    const T poison = gm - gm;   // NaN for rows holding a NaN or an infinity (exp_ clamps)
    T kv[4 * NB];               // w_i of this lane's training points
    T mu = T(0);
    T ga[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ga[d] = T(0);
    // (wide kernels: two training points at a time, their serial chains interleave)
    constexpr int GP = G::kGroup;
    static_assert((4 * NB) % GP == 0, "group size must divide the points per lane");
    static_for<4 * NB / GP>([&](auto qc) {
      constexpr int q0 = decltype(qc)::value * GP;
      T x[GP][D];
      T al[GP], k[GP];
#pragma unroll
      for (int u = 0; u < GP; ++u) {
        const int q = q0 + u;
        const int i = own_index<T>(q >> 2, q & 3, g);
        const T* row = &s_xa[i * DS];
#pragma unroll
        for (int d = 0; d < D; ++d) x[u][d] = row[d];
        al[u] = row[D];
        if constexpr (R::kExpand) k[u] = row[D + 1] + gm;
      }
      // dimension by dimension across the group: GP dependency chains side by side
      if constexpr (R::kExpand) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
          for (int u = 0; u < GP; ++u) k[u] = fma(x[u][d], t[d], k[u]);
        }
      } else {
        T r2[GP];
#pragma unroll
        for (int u = 0; u < GP; ++u) r2[u] = T(0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
          for (int u = 0; u < GP; ++u) {
            const T dl = x[u][d] - t[d];
            r2[u] = fma(dl, dl, r2[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < GP; ++u) k[u] = T(-0.5) * r2[u];
      }
      R::template exp_n<GP>(k);
      if constexpr (!R::kExpand) {
#pragma unroll
        for (int u = 0; u < GP; ++u) k[u] *= b;
      }
#pragma unroll
      for (int u = 0; u < GP; ++u) {
        const T w = k[u] * al[u];
        kv[q0 + u] = w;
        mu += w;
#pragma unroll
        for (int d = 0; d < D; ++d) ga[d] = fma(w, x[u][d], ga[d]);
      }
    });
    mu = xor_reduce_groups(mu) + poison;
    static_for<(D + 5) / 6>([&](auto bc) {
      constexpr int d0 = decltype(bc)::value * 6;
      constexpr int nb_ = (D - d0 < 6) ? (D - d0) : 6;
      xor_reduce_groups_n<T, nb_>(&ga[d0]);
    });
    // The epilogue needs t''_d, G_d for a compile-time d (read back from the wave's LDS slot,
    // which frees 4 D registers during the matrix phase; keeping them in registers in the wide
    // kernels measured 0.7 % slower) and for d2 = 4 bj + g (kept per lane).
    // Lane group g writes the dimensions d = g (mod 4) -- after the reduction all four agree.
    T tq[NB4], gq[NB4];
#pragma unroll
    for (int d = 0; d < D; ++d)
      if ((d & 3) == g) {
        s_tg[wave][ml][d] = t[d];
        s_tg[wave][ml][D + d] = ga[d];
        tq[d >> 2] = t[d];
        gq[d >> 2] = ga[d];
      }
    if constexpr (D % 4 != 0)
      if (4 * (NB4 - 1) + g >= D) tq[NB4 - 1] = gq[NB4 - 1] = T(0);

    // ---------------- phase B: S2 on the matrix core, finish, store ---------------------
    T* out = p.hess + mc * (long long)p.d_actual * p.d_actual;
    const bool row_ok = m < p.M;
    // 16-byte stores need rows that are whole vectors and an aligned matrix (uniform condition).
    // The standard-geometry fp64 instances that are over their 256 registers keep 8-byte stores
    // (the transposed copy costs them more in spills than the wider stores give back); the large
    // ones run wide and store 16 bytes like everything else.
    constexpr bool kVecStores = kWide || !(sizeof(T) == 8 && D > 12 && NB > 12);
    const bool vec_ok = kVecStores && (p.d_actual % (16 / (int)sizeof(T)) == 0) &&
                        (((unsigned long long)p.hess & 15) == 0);
    // Finish and store block `cbv` from its accumulator: acc[r] = S2[d][d2] for d = 4 bi + r,
    // d2 = 4 bj + g, test row ml.  Diagonal blocks hold both (d, d2) and (d2, d); only d <= d2 is
    // used, and mirrored, so that the stored matrix is exactly symmetric.
    auto finish_block = [&](auto cbc, const acc_t& a) {
      constexpr int cbv = decltype(cbc)::value;
      constexpr int bi = hess_block_bi(cbv), bj = hess_block_bj(cbv);
      const int d2 = 4 * bj + g;
#if GP_HESSM_ABLATE == 2   // diagnostic: no epilogue at all
      if (a[0] + a[1] + a[2] + a[3] == T(-12345.678)) out[0] = a[0];
#else
      T v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 4 * bi + r;
        v[r] = T(0);
        if (d < D) {            // (compile time after unrolling)
          const T td = s_tg[wave][ml][d], gd = s_tg[wave][ml][D + d];
          T x = a[r];
          x = fma(-gd, tq[bj], x);
          x = fma(-td, gq[bj], x);
          x = fma(mu * td, tq[bj], x);
          const T sdd = s_sd[d];
          x *= sdd * sdq[bj];
          if (bi == bj && r == g) x = fma(-(sdd * sdd), mu, x);
          v[r] = x;
        }
      }
#if GP_HESSM_ABLATE == 1   // diagnostic: no stores
      if (v[0] + v[1] + v[2] + v[3] == T(-12345.678)) out[0] = v[0];
#else
      if (kVecStores && vec_ok) {
        // 16-byte stores.  The lane's four values are H[4 bi + r][d2], r = 0..3: as the MIRROR
        // image they are four consecutive elements of row d2.  Transposed across the four lane
        // groups (two permlane swaps per register pair) they become four consecutive elements of
        // row 4 bi + g of the block itself.  On a diagonal block only the transposed copy is
        // needed, and each lane keeps the upper-triangle version of every element, so the
        // stored matrix is exactly symmetric.
        T vt[4] = {v[0], v[1], v[2], v[3]};
        transpose_groups4(vt);
        if constexpr (bi != bj) {
          if (row_ok && d2 < p.d_actual) store_row4<T>(out + d2 * p.d_actual + 4 * bi, v, p.d_actual - 4 * bi);
          if (row_ok && 4 * bi + g < p.d_actual)
            store_row4<T>(out + (4 * bi + g) * p.d_actual + 4 * bj, vt, p.d_actual - 4 * bj);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) vt[r] = (r <= g) ? v[r] : vt[r];
          if (row_ok && d2 < p.d_actual) store_row4<T>(out + d2 * p.d_actual + 4 * bi, vt, p.d_actual - 4 * bi);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 4 * bi + r;
          if (row_ok && d <= d2 && d2 < p.d_actual) {   // (d < D <= ... : d2 bounds d)
            out[d * p.d_actual + d2] = v[r];
            if (d != d2) out[d2 * p.d_actual + d] = v[r];
          }
        }
      }
#endif
#endif
    };
    // Blocks are worked in groups -- one block (standard) or a pair whose fragments alternate
    // (wide: two independent accumulator chains) -- on alternating accumulators, and a group is
    // finished a few matrix instructions into the NEXT group: its last instruction has retired
    // by then and the epilogue's VALU and LDS work issues between matrix instructions instead of
    // behind a drained pipe.
    constexpr int kAccs = G::kAccs;
    constexpr int kPer = kWide ? 2 : 1;                 // blocks per group
    constexpr int NGRP = (NBLK + kPer - 1) / kPer;
    acc_t accs[kAccs];
    constexpr int kLag = (NB > 2) ? 2 : NB - 1;         // training block of the next group
    constexpr int kAhead = GP_AHEAD;
    T afr[kAhead];
    auto finish_group = [&](auto gc) {
      constexpr int gv = decltype(gc)::value;
      static_for<kPer>([&](auto jc) {
        constexpr int cbv = gv * kPer + decltype(jc)::value;
        if constexpr (cbv < NBLK) finish_block(std::integral_constant<int, cbv>{}, accs[cbv % kAccs]);
      });
    };
    static_for<NF>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      constexpr int ch = f / kChunk, fl = f % kChunk;
      constexpr HessFragId fid = hess_frag_at(f, NB, kWide, NBLK);
      constexpr int cb = fid.c, I = fid.I, s = fid.s;
      if constexpr (fl == 0) {
        dma_wait();       // this wave's pieces of chunk ch have landed
        __syncthreads();  // chunk ch visible; everyone finished reading chunk ch-1
        if constexpr (ch + 1 < NCH)
          stage_chunk<T, kWaves, kChunk>(p.pfrags + (ch + 1) * kChunk * 64, &s_fr[(ch + 1) & 1][0], wave, lane);
        static_for<kAhead - 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (j < kChunk && f + j < NF) afr[j % kAhead] = s_fr[ch & 1][j * 64 + lane];
        });
      }
      if constexpr (fl + kAhead - 1 < kChunk && f + kAhead - 1 < NF)
        afr[(fl + kAhead - 1) % kAhead] = s_fr[ch & 1][(fl + kAhead - 1) * 64 + lane];
      if constexpr (I == 0 && s == 0) accs[cb % kAccs] = acc_t{T(0), T(0), T(0), T(0)};
#if GP_HESSM_ABLATE == 3   // diagnostic: no matrix instructions (one fma keeps the operands alive)
      if constexpr (s == 0 && I == 0) accs[cb % kAccs][0] = fma(afr[fl % kAhead], kv[4 * I + s], accs[cb % kAccs][0]);
#else
      accs[cb % kAccs] = R::mfma(afr[fl % kAhead], kv[4 * I + s], accs[cb % kAccs]);
#endif
      if constexpr (cb % kPer == 0 && cb >= kPer && I == kLag && s == 0)
        finish_group(std::integral_constant<int, cb / kPer - 1>{});
    });
    finish_group(std::integral_constant<int, NGRP - 1>{});
  }
}

}  // namespace gpk
