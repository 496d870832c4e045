// Fused GP predict kernel for gfx950 (MI355X): mean + variance + gradient in one pass.
//
// Replaces the reference's whole device pipeline -- gp_emulator/gpu/predict.cu:44-155
// (compute_distance, compute_result, compute_error, compute_deriv) and the eight
// kernel_*.cu element-wise kernels + four cuBLAS calls it strings together -- with ONE
// kernel that never materialises an N_train x N_test matrix.  The arithmetic is that of
// the numpy path gp_emulator/GaussianProcess.py:230-247.
//
// Work decomposition (CDNA4, wave64):
//   * a wave owns a tile of 16 test rows; lane l works for test row (l & 15) and for the
//     quarter (l >> 4) of the training points, so the 16 x NP kernel-row tile K_* lives
//     entirely in registers (NP/4 values per lane) in exactly the lane layout the matrix
//     core wants for its B operand AND for its C/D result rows (no LDS round trip);
//   * phase A (VALU): K_* tile, mean and gradient partial sums; training inputs
//     (pre-scaled by sqrt(e_d), as the reference does before cdist) and alpha = invQt are
//     staged once per workgroup in LDS and read as broadcasts;
//   * phase B (MFMA): var = b - k^T invQ k as 16x16x4 MFMAs.  invQ arrives pre-packed in
//     "fragment order" (one 64-lane A operand = 64 consecutive reals) and is streamed
//     L2 -> LDS in double-buffered chunks shared by the 4 waves of the workgroup.  Only
//     block pairs I <= J are visited: S'_IJ = invQ_IJ + invQ_JI^T (I < J), S'_JJ = invQ_JJ,
//     an identity valid for ANY matrix (tests/benchmark.py:14 feeds a non-symmetric one).
//   * 4 waves per workgroup (one per SIMD), 2 workgroups per CU: while one workgroup is in
//     its MFMA phase the other runs its VALU phase on the same SIMDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace gpk {

constexpr int kWaves = 4;            // waves per workgroup (one per SIMD)
constexpr int kThreads = kWaves * 64;
constexpr int kTile = 16;            // test rows per wave tile (MFMA N dimension)
constexpr int kRowsPerWG = kWaves * kTile;
constexpr int kChunk = 32;           // A-operand fragments per LDS chunk

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Compile-time loop: f(std::integral_constant<int, i>) for i in [0, N).  Every register
// array below (kv[], dl[], ga[]) is indexed only through these constants, so nothing is
// runtime-indexed and nothing lands in scratch (cdna_hip_programming.md 5.4 rule 20).
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <typename T> struct Real;
template <> struct Real<double> {
  typedef f64x4 acc_t;
  // row (within a 16-block) of C/D register r for lane group g:
  // v_mfma_f64_16x16x4_f64: row = g + 4 r   (cdna_hip_programming.md section 3)
  __host__ __device__ static constexpr int own_sub(int r, int g) { return 4 * r + g; }
  __device__ static inline acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  __device__ static inline double exp_(double x) { return exp(x); }
};
template <> struct Real<float> {
  typedef f32x4 acc_t;
  // v_mfma_f32_16x16x4_f32: row = 4 g + r
  __host__ __device__ static constexpr int own_sub(int r, int g) { return 4 * g + r; }
  __device__ static inline acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  __device__ static inline float exp_(float x) { return __expf(x); }
};

// Training index that lane group g pairs with k-step s of 16-block I.
template <typename T>
__host__ __device__ constexpr int own_index(int I, int s, int g) {
  return 16 * I + Real<T>::own_sub(s, g);
}

// Position of fragment (I <= J, s) in consumption order.
__host__ __device__ constexpr int frag_index(int I, int J, int s) {
  return (J * (J + 1) / 2 + I) * 4 + s;
}
__host__ __device__ constexpr int frag_count(int NB) { return NB * (NB + 1) / 2 * 4; }
// Inverse of frag_index on the pair index: column block J of pair (I <= J).
__host__ __device__ constexpr int pair_col(int pair) {
  int J = 0;
  while ((J + 1) * (J + 2) / 2 <= pair) ++J;
  return J;
}
// The packed fragment buffer is padded to whole chunks so staging needs no bounds checks.
__host__ __device__ constexpr int frag_count_padded(int NB, int chunk) {
  return (frag_count(NB) + chunk - 1) / chunk * chunk;
}

// Row stride (in reals) of the LDS/global image of [x'_0 .. x'_{D-1}, alpha]: even, so
// rows stay 16-byte aligned for f64.
__host__ __device__ constexpr int row_stride(int D) { return (D + 1 + 3) & ~3; }

template <typename T>
struct PredictArgs {
  const T* xa;        // [16*NB][row_stride(D)]  pre-scaled inputs + alpha (zero padded)
  const T* frags;     // [frag_count_padded(NB,kChunk)][64]  S' in fragment order
  const T* sd;        // [D] sqrt(e_d)
  T b;                // e[D]
  const T* testing;   // [M][d_actual] row-major test inputs (device)
  T* mu;              // [M]
  T* var;             // [M]
  T* deriv;           // [D_actual*M] (d-major, reference layout) or [M][D_actual]
  long long M;
  int d_actual;       // D of the caller (<= template D; extra template dims are zero)
  int deriv_row_major;
};

template <typename T>
__device__ inline T xor_reduce_groups(T v) {
  // sum over the four 16-lane groups (lanes l, l^16, l^32, l^48)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// One chunk (kChunk fragments of 64 reals) global -> LDS with global_load_lds_dwordx4:
// each wave-instruction moves 1 KiB to a wave-uniform LDS base + lane*16, so the linear
// fragment order of the packed buffer is also the LDS order (no swizzle, and the 64-lane
// ds_read_b64 / ds_read_b32 of a fragment is conflict-free).  Completion is covered by the
// vmcnt(0) that __syncthreads() emits while an LDS-DMA is pending (guide section 5).
template <typename T>
__device__ __forceinline__ void stage_chunk(const T* src, T* dst, int wave, int lane) {
  constexpr int kBytes = kChunk * 64 * (int)sizeof(T);
  constexpr int kPerWave = kBytes / kWaves;       // bytes each wave moves
  constexpr int kIters = kPerWave / 1024;
  static_assert(kIters * 1024 * kWaves == kBytes, "chunk must be whole 1 KiB pieces per wave");
  // wave-uniform SGPR base + 32-bit per-lane VGPR offset: the saddr form of the
  // instruction, so no 64-bit per-lane addresses are kept (or hoisted and spilled)
  const char* s = reinterpret_cast<const char*>(src) + wave * kPerWave;
  char* d = reinterpret_cast<char*>(dst) + wave * kPerWave;
  unsigned voff = (unsigned)lane * 16u;
  // opaque re-definition: keeps the (SGPR base + zext VGPR) add next to the instruction so
  // instruction selection sees it, instead of a hoisted 64-bit per-lane pointer
  asm volatile("" : "+v"(voff));
  // the instruction's immediate offset applies to BOTH the global and the LDS address
  static_for<kIters>([&](auto ic) {
    constexpr int it = decltype(ic)::value;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(s + voff),
        (__attribute__((address_space(3))) void*)d, 16, it * 1024, 0);
  });
}

template <typename T, int D, int NB>
__global__ __launch_bounds__(kThreads, 2) void predict_kernel(PredictArgs<T> p) {
  typedef Real<T> R;
  typedef typename R::acc_t acc_t;
  constexpr int NP = 16 * NB;
  constexpr int DS = row_stride(D);
  constexpr int NF = frag_count(NB);
  constexpr int NCH = (NF + kChunk - 1) / kChunk;

  __shared__ __attribute__((aligned(16))) T s_xa[NP * DS];
  __shared__ __attribute__((aligned(16))) T s_fr[2][kChunk * 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int ml = lane & 15;
  const int g = lane >> 4;

  for (int i = tid; i < NP * DS; i += kThreads) s_xa[i] = p.xa[i];
  T sd[D];
#pragma unroll
  for (int d = 0; d < D; ++d) sd[d] = (d < p.d_actual) ? p.sd[d] : T(0);
  const T b = p.b;
  __syncthreads();

  const long long n_groups = (p.M + kRowsPerWG - 1) / kRowsPerWG;
  for (long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long long m = grp * kRowsPerWG + wave * kTile + ml;
    const long long mc = m < p.M ? m : p.M - 1;

    // chunk 0 of S' goes L2 -> LDS by LDS-DMA now and lands under phase A
    __syncthreads();  // previous group's readers of s_fr[0] are done
    stage_chunk<T>(p.frags, &s_fr[0][0], wave, lane);

    // ---------------- phase A: K_* tile, mean, gradient --------------------
    T t[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
      t[d] = (d < p.d_actual) ? sd[d] * p.testing[mc * p.d_actual + d] : T(0);

    T kv[4 * NB];
    T mu = T(0);
    T ga[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ga[d] = T(0);

    static_for<4 * NB>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int i = own_index<T>(q >> 2, q & 3, g);
      const T* row = &s_xa[i * DS];
      T dl[D];
      T r2 = T(0);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        dl[d] = row[d] - t[d];
        r2 = fma(dl[d], dl[d], r2);
      }
      const T k = b * R::exp_(T(-0.5) * r2);
      kv[q] = k;
      const T w = k * row[D];
      mu += w;
#pragma unroll
      for (int d = 0; d < D; ++d) ga[d] = fma(w, dl[d], ga[d]);
    });
    mu = xor_reduce_groups(mu);
#pragma unroll
    for (int d = 0; d < D; ++d) ga[d] = xor_reduce_groups(ga[d]) * sd[d];

    if (m < p.M) {
      if (g == 0) p.mu[m] = mu;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if ((d & 3) == g && d < p.d_actual) {
          if (p.deriv_row_major) p.deriv[m * p.d_actual + d] = ga[d];
          else p.deriv[(long long)d * p.M + m] = ga[d];
        }
      }
    }

    // ---------------- phase B: variance on the matrix core -----------------
    T vacc = T(0);
    acc_t acc;
    static_for<NF>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      constexpr int c = f / kChunk, fl = f % kChunk;
      constexpr int pair = f >> 2, s = f & 3;
      constexpr int J = pair_col(pair);
      constexpr int I = pair - J * (J + 1) / 2;
      if constexpr (fl == 0) {
        __syncthreads();  // chunk c visible; everyone finished reading chunk c-1
        if constexpr (c + 1 < NCH)
          stage_chunk<T>(p.frags + (c + 1) * kChunk * 64, &s_fr[(c + 1) & 1][0], wave, lane);
      }
      if constexpr (I == 0 && s == 0) acc = acc_t{T(0), T(0), T(0), T(0)};
      acc = R::mfma(s_fr[c & 1][fl * 64 + lane], kv[4 * I + s], acc);
      if constexpr (I == J && s == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vacc = fma(acc[r], kv[4 * J + r], vacc);
      }
    });
    vacc = xor_reduce_groups(vacc);
    if (m < p.M && g == 1) p.var[m] = b - vacc;
  }
}

}  // namespace gpk
