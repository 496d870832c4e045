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
//     L2 -> LDS by LDS-DMA in double-buffered chunks shared by the waves of the workgroup.
//     Only block pairs I >= J are visited: S'_IJ = invQ_IJ + invQ_JI^T (I > J), S'_JJ =
//     invQ_JJ, an identity valid for ANY matrix (tests/benchmark.py:14 feeds a non-symmetric
//     one).  The contraction runs over the training blocks I, and the kernel is compiled for
//     a number of k-steps NK (groups of 4 training points), not of 16-blocks: the k-steps of
//     the last block that would hold nothing but padding (N = 250 -> NK = 63: rows 252..255)
//     are issued neither in phase A nor, for any column block J, in phase B;
//   * geometry (struct Geo): fp64 8 waves per workgroup = two per SIMD, one workgroup per CU,
//     all waves in the same phase (fp64 MFMA and fp64 VALU share one pipe, so there is
//     nothing to overlap); fp32 12 waves = three per SIMD.  Persistent grid, one workgroup
//     per CU, work items dealt round-robin.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace gpk {

#ifndef GP_WAVES
#define GP_WAVES 8
#endif
#ifndef GP_WAVES_F32
#define GP_WAVES_F32 12
#endif
#ifndef GP_CHUNK
#define GP_CHUNK 64
#endif
#ifndef GP_CHUNK_F32
#define GP_CHUNK_F32 192   // fp32: 96 KB of fragment buffers, three chunks at N = 250 (round 3: 128 -> 192, -3 %; 64: +0 %)
#endif
// Timing-only ablations for tools/ab_bench.py (outputs are wrong when set): GP_ABLATE=1 skips
// the matrix-core phase, GP_ABLATE=2 replaces phase A's arithmetic by a trivial fill, GP_ABLATE=3 takes the matrix
// instructions' A operands from registers instead of the LDS fragments (barriers and DMA stay).
#ifndef GP_ABLATE
#define GP_ABLATE 0
#endif
#ifndef GP_AHEAD
// matrix-core phase: A-operand LDS reads in flight (a ring of registers).  A/B on one device
// (profiles/r02_ab_kernel_variants.txt): 1 (read, wait, issue) 1.452 ms, 2: 1.443, 4: 1.435,
// 6: 1.435; folding a column block into the variance sum two instructions into the NEXT block
// (second accumulator) on top of 4: 1.442 -- not kept.  fp32: no difference.
#define GP_AHEAD 4
#endif
#ifndef GP_AHEAD_F32
#define GP_AHEAD_F32 8    // fp32, with the order pinned (GP_RING_PIN): 4 -> 8 reads in flight, -1.5 %
#endif
#ifndef GP_RING_PIN
#define GP_RING_PIN 1     // 1: fp32, 2: both precisions, 0: off (see the matrix-core phase)
#endif
#ifndef GP_MU_DOUBLE
#define GP_MU_DOUBLE 1    // 0: the mean's sum in the compute type (A/B: fp32 accuracy on ill-conditioned emulators, speed)
#endif
#ifndef GP_ESTRIN
#define GP_ESTRIN 0     // 1: Estrin form of the exp polynomial (A/B: slower, more registers)
#endif
#ifndef GP_DOTSPLIT
#define GP_DOTSPLIT 1     // independent partial sums of the exponent's dot product
#endif
#ifndef GP_GROUP_F64
#define GP_GROUP_F64 1
#endif
// GP_STAMPS=1: diagnostic build (tools/stamp_profile.py).  Every wave accumulates the shader
// cycles it spends in six segments of an item and adds them to PredictArgs::dbg at the end.
// Never used for timing or results (cdna_hip_programming.md section 7, in-kernel stamps).
#ifndef GP_STAMPS
#define GP_STAMPS 0
#endif
// Predict kernel geometry, per compute type.  fp64: 8 waves = two per SIMD (the register
// budget of the K_* tile), all in the same phase: fp64 MFMA and fp64 VALU share one pipe on
// MI355X (tools/mfma_f64_probe.hip), so there is nothing to gain from running a VALU-phase
// workgroup beside an MFMA-phase one, and partners in the same phase keep the workgroup's
// waves in step (short barrier waits) while each fragment staged in LDS feeds every wave.
constexpr int kTile = 16;            // test rows per wave tile (MFMA N dimension)
template <typename T> struct Geo {
  static constexpr int kWaves = sizeof(T) == 8 ? GP_WAVES : GP_WAVES_F32;  // per workgroup
  static constexpr int kThreads = kWaves * 64;
#ifndef GP_WG_PER_CU_SMALL
#define GP_WG_PER_CU_SMALL 0    // > 0: workgroups per CU when a workgroup is four waves (one per SIMD), else 8 / waves
#endif
  static constexpr int kWGPerCU = kWaves >= 8 ? 1 : (GP_WG_PER_CU_SMALL > 0 ? GP_WG_PER_CU_SMALL : 8 / kWaves);
  static constexpr int kWavesPerSimd = kWaves >= 8 ? kWaves / 4 : kWGPerCU * kWaves / 4;  // launch bound
  static constexpr int kRowsPerWG = kWaves * kTile;
  // A-operand fragments per LDS chunk (double-buffered: 2 x kChunk x 64 reals of LDS)
  static constexpr int kChunk = sizeof(T) == 8 ? GP_CHUNK : GP_CHUNK_F32;
};

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
  // phase A in expansion form (k = exp(h_i + g + x''.t''), 2 fma per point and dimension):
  // the cancellation costs ~|x''|^2 ulps of 1e-16, far inside the 1e-10 fp64 bar
  static constexpr bool kExpand = true;
  static constexpr int kGroup = GP_GROUP_F64;   // training points per software-pipelined group
  // row (within a 16-block) of C/D register r for lane group g:
  // v_mfma_f64_16x16x4_f64: row = g + 4 r   (cdna_hip_programming.md section 3)
  __host__ __device__ static constexpr int own_sub(int r, int g) { return 4 * r + g; }
  __device__ static inline acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // exp for arguments that are <= 0 up to rounding (-0.5 * squared distance + ln b):
  // n = rint(x / ln2), r = x - n ln2 (two-step, fma), exp(r) by the degree-10 Chebyshev
  // interpolant of exp on |r| <= ln2/2 (max relative error 3.3e-16, coefficients derived in
  // 80-bit arithmetic), scaled by v_ldexp_f64, which also delivers the gradual underflow to
  // 0.  No overflow path is needed.  The lower clamp (one v_max_f64) keeps n inside int32
  // for absurdly distant points; it would turn a NaN argument into exp(-1000) = 0, so the
  // caller adds the NaN back per test row (see `poison` in predict_kernel).
  __device__ static inline double exp_(double x) {
    x = __builtin_fmax(x, -1000.0);
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = fma(n, -6.93147180559945286e-01, x);
    r = fma(n, -2.31904681384629956e-17, r);
#if GP_ESTRIN
    // Estrin evaluation: 13 instructions instead of Horner's 10, but a dependent chain of 5
    // instead of 10 -- phase A is bound by fp64 dependent-issue latency (~20 cycles per
    // dependent instruction, one chain per wave), not by instruction count
    const double r2 = r * r;
    const double p01 = fma(1.00000000000000666e+00, r, 1.0);
    const double p23 = fma(1.66666666665544028e-01, r, 5.00000000000000555e-01);
    const double p45 = fma(8.33333338566834801e-03, r, 4.16666666665732183e-02);
    const double p67 = fma(1.98411702695461072e-04, r, 1.38888889324666632e-03);
    const double p89 = fma(2.76401812311866076e-06, r, 2.48015043709117912e-05);
    const double r4 = r2 * r2;
    const double q0 = fma(r2, p23, p01);
    const double q1 = fma(r2, p67, p45);
    const double q2 = fma(r2, 2.76263485910095559e-07, p89);
    const double r8 = r4 * r4;
    const double s0 = fma(r4, q1, q0);
    const double p = fma(r8, q2, s0);
#else
    double p = 2.76263485910095559e-07;
    p = fma(p, r, 2.76401812311866076e-06);
    p = fma(p, r, 2.48015043709117912e-05);
    p = fma(p, r, 1.98411702695461072e-04);
    p = fma(p, r, 1.38888889324666632e-03);
    p = fma(p, r, 8.33333338566834801e-03);
    p = fma(p, r, 4.16666666665732183e-02);
    p = fma(p, r, 1.66666666665544028e-01);
    p = fma(p, r, 5.00000000000000555e-01);
    p = fma(p, r, 1.00000000000000666e+00);
    p = fma(p, r, 1.0);
#endif
    return ldexp(p, (int)n);
  }
  // exp_ of GP values, written step by step across the values: GP independent dependency chains
  // side by side in program order (for kernels that run one wave per SIMD, where nothing else
  // hides the latency of a dependent fp64 instruction)
  template <int GP>
  __device__ static inline void exp_n(double (&x)[GP]) {
    double n[GP], r[GP], p[GP];
#pragma unroll
    for (int u = 0; u < GP; ++u) x[u] = __builtin_fmax(x[u], -1000.0);
#pragma unroll
    for (int u = 0; u < GP; ++u) n[u] = __builtin_rint(x[u] * 1.4426950408889634);
#pragma unroll
    for (int u = 0; u < GP; ++u) r[u] = fma(n[u], -6.93147180559945286e-01, x[u]);
#pragma unroll
    for (int u = 0; u < GP; ++u) r[u] = fma(n[u], -2.31904681384629956e-17, r[u]);
#pragma unroll
    for (int u = 0; u < GP; ++u) p[u] = 2.76263485910095559e-07;
    constexpr double c[10] = {2.76401812311866076e-06, 2.48015043709117912e-05, 1.98411702695461072e-04,
                              1.38888889324666632e-03, 8.33333338566834801e-03, 4.16666666665732183e-02,
                              1.66666666665544028e-01, 5.00000000000000555e-01, 1.00000000000000666e+00, 1.0};
#pragma unroll
    for (int j = 0; j < 10; ++j) {
#pragma unroll
      for (int u = 0; u < GP; ++u) p[u] = fma(p[u], r[u], c[j]);
    }
#pragma unroll
    for (int u = 0; u < GP; ++u) x[u] = ldexp(p[u], (int)n[u]);
  }
};
template <> struct Real<float> {
  typedef f32x4 acc_t;
  // phase A in difference form (k = b exp(-|x''-t''|^2/2), 3 ops per point and dimension):
  // in fp32 the expansion's cancellation would cost ~4x accuracy on ill-conditioned real
  // emulators, and fp32 VALU ops are half the price of fp64 ones
  static constexpr bool kExpand = false;
#ifndef GP_GROUP_F32
#define GP_GROUP_F32 2
#endif
  static constexpr int kGroup = GP_GROUP_F32;
  // v_mfma_f32_16x16x4_f32: row = 4 g + r
  __host__ __device__ static constexpr int own_sub(int r, int g) { return 4 * g + r; }
  __device__ static inline acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  __device__ static inline float exp_(float x) { return __expf(x); }
  template <int GP>
  __device__ static inline void exp_n(float (&x)[GP]) {
#pragma unroll
    for (int u = 0; u < GP; ++u) x[u] = __expf(x[u]);
  }
};

// Training index that lane group g pairs with k-step s of 16-block I.
template <typename T>
__host__ __device__ constexpr int own_index(int I, int s, int g) {
  return 16 * I + Real<T>::own_sub(s, g);
}

// Slot of training point n in the packed images (rows of xa, rows and columns of S'): inside its
// 16-block the n'-th point goes to the row that k-step n' / 4 pairs with lane group n' % 4, so
// the points of a partly filled last block fill whole k-steps first (fp64: the identity; fp32,
// whose matrix-core rows are 4 g + r: a 4 x 4 transpose).  slot_point is the inverse.
template <typename T>
__host__ __device__ constexpr int point_slot(int n) {
  return 16 * (n / 16) + Real<T>::own_sub((n % 16) / 4, n % 4);
}
template <typename T>
__host__ __device__ constexpr int slot_point(int slot) {
  for (int q = 0; q < 16; ++q)
    if (Real<T>::own_sub(q / 4, q % 4) == slot % 16) return 16 * (slot / 16) + q;
  return -1;
}

// Fragment order: column block by column block, (J, I >= J, s).  Fragment (I, J, s) is the
// A operand of the MFMA that adds training block I's k-step s to column block J's accumulator:
// lane l holds S'[16 I + own_sub(s, l >> 4)][16 J + (l & 15)].
struct FragId { int I, J, s; };
__host__ __device__ constexpr int frag_count(int NB) { return NB * (NB + 1) / 2 * 4; }
// block pairs in front of column block J
__host__ __device__ constexpr int frag_col_offset(int J, int NB) { return J * NB - J * (J - 1) / 2; }
// n-th fragment in consumption order (evaluated at compile time in the kernel)
__host__ __device__ constexpr FragId frag_at(int n, int NB) {
  const int pair = n >> 2;
  int J = 0;
  while (J + 1 < NB && frag_col_offset(J + 1, NB) <= pair) ++J;
  return FragId{J + pair - frag_col_offset(J, NB), J, n & 3};
}
// position of fragment (I >= J, s) in consumption order (host-side packing and tests)
__host__ __device__ constexpr int frag_index(int I, int J, int s, int NB) {
  return (frag_col_offset(J, NB) + (I - J)) * 4 + s;
}
// The packed fragment buffer is padded to whole chunks so staging needs no bounds checks.
__host__ __device__ constexpr int frag_count_padded(int NB, int chunk) {
  return (frag_count(NB) + chunk - 1) / chunk * chunk;
}

// Row stride (in reals) of the LDS/global image of one training point,
//   [x''_0 .. x''_{D-1}, alpha_i, h_i],   x'' = sqrt(e) (x - c),  h_i = ln b - |x''_i|^2 / 2,
// rounded up to a multiple of 4 reals so rows stay 16-byte aligned in both precisions.
__host__ __device__ constexpr int row_stride(int D) { return (D + 2 + 3) & ~3; }
// Row stride of predict_kernel's LDS copy.  Phase A reads a row with ds_read_b128, which the LDS serves 16 lanes at a
// time in groups that mix two of the wave's four lane groups (MI355X_MICROARCH.md, LDS: lanes 0-3, 12-15 with 20-27,
// ...): two different rows per LDS cycle.  fp64 pairs adjacent rows (own_sub = 4 r + g), 128 bytes apart at D = 11:
// opposite halves of the 256-byte bank row.  fp32 pairs rows FOUR apart (own_sub = 4 g + r): with 64-byte rows that
// is 256 bytes, the same banks, and every read took two cycles per group (SQ_LDS_BANK_CONFLICT 29 % of the kernel's
// LDS cycles, profiles/r03_fp32_lds_conflicts.txt) -- one more 16-byte piece per row moves the partner 64 bytes on.
#ifndef GP_XA_PAD
#define GP_XA_PAD 1
#endif
template <typename T> __host__ __device__ constexpr int xa_lds_stride(int D) {
  return (GP_XA_PAD && sizeof(T) == 4 && row_stride(D) % 16 == 0) ? row_stride(D) + 4 : row_stride(D);
}

template <typename T>
struct PredictArgs {
  const T* xa;        // [16*NB][row_stride(D)]  training rows [x'', alpha, h] (zero padded), NB = ceil(NK/4)
  const T* frags;     // [frag_count_padded(NB,kChunk)][64]  S' in fragment order
  const T* sd;        // [2*D + 1] sqrt(e_d), the centre c_d (training mean, input units), b = e[D]
  const T* testing;   // [M][d_actual] row-major test inputs (device)
  T* mu;              // [M]
  T* var;             // [M]
  T* deriv;           // [D_actual*M] (d-major, reference layout) or [M][D_actual]
  long long M;
  int d_actual;       // D of the caller (<= template D; extra template dims are zero)
  int deriv_row_major;
  // Batched emulators (per-band pattern, tests/test_perband_emulator.py:22-37): E emulators
  // share the test rows; emulator e has its constants at xa + e*xa_stride,
  // frags + e*frags_stride, sd + e*sd_stride and its outputs at mu + e*M, var + e*M,
  // deriv + e*M*d_actual.  n_emulators = 1 is the plain single-emulator call.
  int n_emulators;
  long long xa_stride, frags_stride, sd_stride;
  unsigned long long* dbg;   // GP_STAMPS builds only: [8] segment cycle sums; else unused
  // 1: `testing` already holds t'' = sqrt(e)(t - c) (scaled and centred in double by the host
  // while staging float64 rows for a float32 predict); 0: raw rows, scaled here
  int rows_prescaled;
};

// v[l] + v[l ^ 16] (XOR = 16) or v[l] + v[l ^ 32] (XOR = 32) in every lane, with gfx950's
// v_permlane16_swap / v_permlane32_swap: swapping a register with a copy of itself leaves
// "every lane holds its own half's value" in one result and "the other half's" in the other.
// Pure VALU -- no LDS crossbar round trip, no s_waitcnt -- and bit-identical to the __shfl_xor
// form (a + b == b + a).
template <int XOR>
__device__ __forceinline__ unsigned swap_halves(unsigned x, unsigned& other) {
  if constexpr (XOR == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    other = r[1];
    return r[0];
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    other = r[1];
    return r[0];
  }
}
template <int XOR>
__device__ __forceinline__ float pair_sum(float v) {
  unsigned o, a = swap_halves<XOR>(__float_as_uint(v), o);
  return __uint_as_float(a) + __uint_as_float(o);
}
template <int XOR>
__device__ __forceinline__ double pair_sum(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  unsigned lo_o, hi_o;
  const unsigned lo_a = swap_halves<XOR>((unsigned)u, lo_o);
  const unsigned hi_a = swap_halves<XOR>((unsigned)(u >> 32), hi_o);
  const double a = __longlong_as_double((long long)(((unsigned long long)hi_a << 32) | lo_a));
  const double o = __longlong_as_double((long long)(((unsigned long long)hi_o << 32) | lo_o));
  return a + o;
}

// Same for N values at once (kept as an interface: with the swap form there is no latency to
// batch, the values are simply reduced one after the other).
template <typename T, int N>
__device__ inline void xor_reduce_groups_n(T* v) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = pair_sum<32>(pair_sum<16>(v[i]));
}

template <typename T>
__device__ inline T xor_reduce_groups(T v) {
  // sum over the four 16-lane groups (lanes l, l^16, l^32, l^48)
  return pair_sum<32>(pair_sum<16>(v));
}

// One chunk (kChunk fragments of 64 reals) global -> LDS by LDS-DMA
// (global_load_lds_dwordx4): each wave-instruction moves 1 KiB to a wave-uniform LDS base
// (M0) + lane*16, so the linear fragment order of the packed buffer is also the LDS order
// (no swizzle; the 64-lane ds_read_b64 / ds_read_b32 of a fragment is conflict-free).
//
// Written as inline asm on purpose: with the builtin, hipcc puts `s_waitcnt vmcnt(0)` in
// front of the next ds_read (it cannot prove the DMA's LDS write does not alias it), which
// serialises the copy with the MFMA work it is meant to hide under.  The contract for the
// asm form (cdna_hip_programming.md 5.7): nothing reads the destination buffer until the
// issuing wave has run dma_wait() and the workgroup has passed a barrier.  Compiler-placed
// vmcnt waits stay safe: un-counted extra operations can only make them wait longer.
// M0 (the DMA's LDS base) is compiler-reserved and not preserved around an asm statement, and
// it cannot be listed as a clobber: the statement that reads it also writes it, and puts the
// previous value back before it ends (cdna_hip_programming.md 5.7, "Operands and clobbers").
template <typename T, int kWaves = Geo<T>::kWaves, int kChunk = Geo<T>::kChunk>
__device__ __forceinline__ void stage_chunk(const T* src, T* dst, int wave, int lane) {
  constexpr int kBytes = kChunk * 64 * (int)sizeof(T);
  constexpr int kPieces = kBytes / 1024;          // 1 KiB per wave-instruction
  static_assert(kPieces * 1024 == kBytes, "chunk must be whole 1 KiB pieces");
  // (round 3, tried: waves 0..3 -- one per SIMD -- issuing all the pieces, the others running the same
  // instructions with EXEC = 0, so that a wave's SIMD partner keeps the matrix pipe busy while it issues:
  // 1.456 vs 1.435 ms on config 2, fp64 -- the four issuing waves then run behind at every chunk barrier)
  // wave-uniform SGPR source base, 32-bit per-lane VGPR offset (saddr form); piece pc goes
  // to wave (pc mod kWaves)
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned lds0 = (unsigned)(uintptr_t)(
      (__attribute__((address_space(3))) char*)(reinterpret_cast<char*>(dst)));
#pragma unroll
  for (int pc0 = 0; pc0 < kPieces; pc0 += kWaves) {
    const int pc = pc0 + wave;
    if (pc0 + kWaves <= kPieces || pc < kPieces) {
      const char* s = reinterpret_cast<const char*>(src) + pc * 1024;
      const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + pc * 1024);
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %1\n\t"
          "s_nop 4\n\t"   /* covers VALU(v_readlane)->SGPR->VMEM and M0->LDS-DMA wait states */
          "global_load_lds_dwordx4 %2, %3\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "s"(m0v), "v"(voff), "s"(s)
          : "memory");
    }
  }
}
// NP consecutive 1 KiB pieces, first_piece .. first_piece + NP - 1 (wave-uniform), by ONE wave: same
// instruction sequence and contract as stage_chunk.
template <typename T, int NP>
__device__ __forceinline__ void stage_pieces(const T* src, T* dst, int first_piece, int lane) {
  const unsigned voff = (unsigned)lane * 16u;
  const unsigned lds0 = (unsigned)(uintptr_t)(
      (__attribute__((address_space(3))) char*)(reinterpret_cast<char*>(dst)));
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pc = first_piece + i;
    const char* s = reinterpret_cast<const char*>(src) + pc * 1024;
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + pc * 1024);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %2, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(m0v), "v"(voff), "s"(s)
        : "memory");
  }
}
// Retire this wave's outstanding LDS-DMA pieces; call right before the barrier that
// publishes the chunk.
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#if GP_STAMPS == 1     // (2: the Hessian kernel's wave lifetimes only -- no stamp inside an item)
#define GP_STAMP(seg)                                                                    \
  do {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    unsigned long long now_;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    seg_sum[seg] += now_ - seg_t0;                                                       \
    seg_t0 = now_;                                                                       \
  } while (0)
#else
#define GP_STAMP(seg) do { } while (0)
#endif

// NK = k-steps (groups of 4 training points) the kernel is compiled for: n_train <= 4 NK.
template <typename T, int D, int NK>
__global__ __launch_bounds__(Geo<T>::kThreads, Geo<T>::kWavesPerSimd) void predict_kernel(PredictArgs<T> p) {
  typedef Real<T> R;
  constexpr int kThreads = Geo<T>::kThreads;
  constexpr int kRowsPerWG = Geo<T>::kRowsPerWG;
  typedef typename R::acc_t acc_t;
  constexpr int NB = (NK + 3) / 4;          // 16-blocks of training points
  constexpr int KL = NK - 4 * (NB - 1);     // live k-steps of the last block, 1..4
  constexpr int NP = 16 * NB;
  constexpr int DSG = row_stride(D);        // the packed image in global memory
  constexpr int DS = xa_lds_stride<T>(D);   // its copy in LDS
  constexpr int NF = frag_count(NB);
  constexpr int kChunk = Geo<T>::kChunk;
  constexpr int NCH = (NF + kChunk - 1) / kChunk;

  __shared__ __attribute__((aligned(16))) T s_xa[NP * DS];
  __shared__ __attribute__((aligned(16))) T s_fr[2][kChunk * 64];
  __shared__ T s_sd[2 * D + 1];   // sqrt(e_d), centre c_d, b: broadcast reads, no registers
  __shared__ T s_ts[2 * D];       // scale and centre applied to the test rows: s_sd's, or (1, 0)
  // raw test rows of the wave's tile, double-buffered and private to the wave: the NEXT item's
  // rows are fetched (coalesced, 8 or 4 B per lane) while the matrix-core phase runs and are
  // picked up from here at the top of that item -- all waves of a workgroup are in step, so
  // nothing else would hide the HBM latency of those loads
  __shared__ T s_rows[2][Geo<T>::kWaves][kTile * D];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int ml = lane & 15;
  const int g = lane >> 4;

  // Work items are (emulator, group of 64 test rows), emulator-major, dealt round-robin to
  // the persistent workgroups: at any moment the whole chip works on one or two emulators,
  // so each XCD's L2 pulls an emulator's S' from HBM once and serves every CU from it.
  // (e, grp) are advanced with scalar adds/compares only, so they stay in SGPRs.
  const int n_groups = (int)((p.M + kRowsPerWG - 1) / kRowsPerWG);
  int e = 0, cur_e = -1;
  int grp = blockIdx.x;
  while (grp >= n_groups && e < p.n_emulators) { grp -= n_groups; ++e; }

  constexpr int NJ = (kTile * D + 63) / 64;        // coalesced loads per lane per tile
  auto fetch_rows = [&](int grp_, T (&regs)[NJ]) {
    const long long e0 = ((long long)grp_ * kRowsPerWG + wave * kTile) * p.d_actual;
    const long long emax = p.M * p.d_actual - 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      long long e_ = e0 + lane + 64 * j;
      e_ = e_ < emax ? e_ : emax;                  // tail tile: stay inside the array
      regs[j] = p.testing[e_];
    }
  };
  auto stash_rows = [&](int buf_, const T (&regs)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (lane + 64 * j < kTile * p.d_actual) s_rows[buf_][wave][lane + 64 * j] = regs[j];
  };
  int rbuf = 0;
  T rregs[NJ];
  if (e < p.n_emulators) {
    fetch_rows(grp, rregs);
    stash_rows(0, rregs);
  }
#if GP_STAMPS
  unsigned long long seg_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long seg_t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg_t0)::"memory");
#endif

  for (; e < p.n_emulators;) {
    if (e != cur_e) {   // (re)load this emulator's training rows and scalars
      cur_e = e;
      __syncthreads();  // everyone is done with the previous emulator's rows
      const T* xa = p.xa + e * p.xa_stride;
      for (int i = tid; i < NP * DSG; i += kThreads) s_xa[(i / DSG) * DS + i % DSG] = xa[i];
      const T* sdp = p.sd + e * p.sd_stride;
      if (tid < 2 * D + 1) {
        const bool live = tid == 2 * D || (tid % D) < p.d_actual;
        s_sd[tid] = live ? sdp[tid] : T(0);
        if (tid < 2 * D)
          s_ts[tid] = !live ? T(0) : !p.rows_prescaled ? sdp[tid] : tid < D ? T(1) : T(0);
      }
    }
    const T* frags = p.frags + e * p.frags_stride;
    T* o_mu = p.mu + e * p.M;
    T* o_var = p.var + e * p.M;
    T* o_der = p.deriv + e * p.M * p.d_actual;
    const long long m = (long long)grp * kRowsPerWG + wave * kTile + ml;
    const long long mc = m < p.M ? m : p.M - 1;

    // chunk 0 of S' goes L2 -> LDS by LDS-DMA now and lands under phase A
    __syncthreads();  // previous item's readers of s_fr[0] are done; new rows visible
    stage_chunk<T>(frags, &s_fr[0][0], wave, lane);
    const T b = s_sd[2 * D];
    GP_STAMP(0);   // emulator switch, barrier, DMA issue

    // ---------------- phase A: K_* tile, mean, gradient --------------------
    // k_i = b exp(-|x''_i - t''|^2 / 2) = exp(h_i + g + x''_i . t''),  g = -|t''|^2 / 2:
    // one fma per (training point, dimension) for the kernel row and one for the gradient
    // sum  G_d = sum_i w_i x''_id,  deriv_d = sqrt(e_d) (G_d - t''_d mu).  Centring on the
    // training mean c keeps |x''|, |t''| (hence the cancellation in h + g + x.t) small.
    T t[D];
    T gm = T(0);
    {
      // all LDS reads first, then the arithmetic: one LDS round trip instead of one per
      // dimension.  Dimensions beyond d_actual have sd = centre = 0 (zero padding) and read a
      // clamped, valid element, so they come out as exactly 0 without a select.
      T sdv[D], cv[D], rv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int dc = d < p.d_actual ? d : p.d_actual - 1;
        sdv[d] = s_ts[d];
        cv[d] = s_ts[D + d];
        rv[d] = s_rows[rbuf][wave][ml * p.d_actual + dc];
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {
        t[d] = sdv[d] * (rv[d] - cv[d]);
        gm = fma(t[d], t[d], gm);
      }
    }
    gm *= T(-0.5);
    // 0 for finite test rows, NaN for rows holding a NaN or an infinity: added to every
    // output of the row so that bad inputs surface as NaN despite the clamp inside exp_
    const T poison = gm - gm;

    asm volatile("" :: "v"(gm));
    // next item (scalar bookkeeping); its test rows are fetched now, in flight during phase A
    int e_next = e, grp_next = grp + gridDim.x;
    while (grp_next >= n_groups && e_next < p.n_emulators) { grp_next -= n_groups; ++e_next; }
    // unconditional (a harmless re-fetch of this item's rows when there is no next item):
    // no branch, so the fetched values' live range stays simple for the register allocator
    fetch_rows(e_next < p.n_emulators ? grp_next : grp, rregs);

    GP_STAMP(1);   // test rows loaded and scaled
    T kv[NK];
    // the mean's sum runs in double in BOTH precisions: on an ill-conditioned emulator (sum |k a| / |mean| ~ 1e5 on
    // PROSAIL) float32 accumulation of the float32 products is what the float32 kernel loses first (mean error
    // 9.7e-4 -> 4.7e-4 of max|mean| with the sum alone in double; what is left is k_i itself, computed in float32);
    // two fp64-rate instructions per training point
#if GP_MU_DOUBLE
    double mu = 0.0;
#else
    T mu = T(0);
#endif
    T ga[D];
#pragma unroll
    for (int d = 0; d < D; ++d) ga[d] = T(0);

    // Training points are handled in groups of R::kGroup whose stages (LDS reads +
    // distance, exp, gradient update) are written out stage by stage, so that within a
    // group the loads are in flight together and the serial distance chains interleave.
    // fp64 has no registers to spare for that (kGroup = 1); fp32 does.
    constexpr int GP = R::kGroup;
    static_for<(NK + GP - 1) / GP>([&](auto qc) {
      constexpr int q0 = decltype(qc)::value * GP;
      constexpr int GU = NK - q0 < GP ? NK - q0 : GP;     // (the last group may be short)
      T x[GP][D];
      T al[GP];
      T k[GP];
      // stage 1: rows -> registers, exponent argument
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int q = q0 + u;
        const int i = own_index<T>(q >> 2, q & 3, g);
        const T* row = &s_xa[i * DS];
#pragma unroll
        for (int d = 0; d < D; ++d) x[u][d] = row[d];
        al[u] = row[D];
        if constexpr (R::kExpand) k[u] = row[D + 1] + gm;
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
#if GP_ABLATE == 2
        continue;
#endif
        if constexpr (R::kExpand) {
          // GP_DOTSPLIT independent partial sums: shortens the dependent fma chain
          T part[GP_DOTSPLIT];
          part[0] = k[u];
#pragma unroll
          for (int c = 1; c < GP_DOTSPLIT; ++c) part[c] = T(0);
#pragma unroll
          for (int d = 0; d < D; ++d) part[d % GP_DOTSPLIT] = fma(x[u][d], t[d], part[d % GP_DOTSPLIT]);
#pragma unroll
          for (int c = 1; c < GP_DOTSPLIT; ++c) part[0] += part[c];
          k[u] = part[0];
        } else {
          T r2 = T(0);
#pragma unroll
          for (int d = 0; d < D; ++d) {
            x[u][d] -= t[d];
            r2 = fma(x[u][d], x[u][d], r2);
          }
          k[u] = T(-0.5) * r2;
        }
      }
      // stage 2: kernel values
#pragma unroll
      for (int u = 0; u < GU; ++u) {
#if GP_ABLATE == 2
        k[u] = x[u][0] + t[0];
#else
        k[u] = R::kExpand ? R::exp_(k[u]) : b * R::exp_(k[u]);
#endif
        kv[q0 + u] = k[u];
      }
      // stage 3: mean and gradient sums
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const T w = k[u] * al[u];
        mu += (decltype(mu))w;
#if GP_ABLATE == 2
        ga[u % D] += w;
#else
#pragma unroll
        for (int d = 0; d < D; ++d) ga[d] = fma(w, x[u][d], ga[d]);
#endif
      }
    });
    GP_STAMP(2);   // phase A proper
    // (batches of 6: a whole-array batch raises the live-register peak enough to spill)
    const T mu_t = (T)xor_reduce_groups(mu) + poison;
    static_for<(D + 5) / 6>([&](auto bc) {
      constexpr int d0 = decltype(bc)::value * 6;
      constexpr int nb_ = (D - d0 < 6) ? (D - d0) : 6;
      xor_reduce_groups_n<T, nb_>(&ga[d0]);
    });
    {
      T sdv[D];
#pragma unroll
      for (int d = 0; d < D; ++d) sdv[d] = s_sd[d];
#pragma unroll
      for (int d = 0; d < D; ++d)
        ga[d] = sdv[d] * (R::kExpand ? fma(-t[d], mu_t, ga[d]) : ga[d]);
    }

    if (m < p.M) {
      if (g == 0) o_mu[m] = mu_t;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if ((d & 3) == g && d < p.d_actual) {
          if (p.deriv_row_major) o_der[m * p.d_actual + d] = ga[d];
          else o_der[(long long)d * p.M + m] = ga[d];
        }
      }
    }

    GP_STAMP(3);   // lane-group reductions, mean / gradient stores
    stash_rows(rbuf ^ 1, rregs);   // next item's rows (fetched during phase A) -> LDS

    // ---------------- phase B: variance on the matrix core -----------------
    T vacc = T(0);
    acc_t acc;
    constexpr int kAhead = sizeof(T) == 4 ? GP_AHEAD_F32 : GP_AHEAD;
    T afr[kAhead];
#if GP_ABLATE == 1
    static_for<NK>([&](auto qc) { vacc += kv[decltype(qc)::value]; });
    static_for<0>([&](auto fc) {
#else
    static_for<NF>([&](auto fc) {
#endif
      constexpr int f = decltype(fc)::value;
      constexpr int c = f / kChunk, fl = f % kChunk;
      constexpr FragId fid = frag_at(f, NB);
      constexpr int I = fid.I, J = fid.J, s = fid.s;
      if constexpr (fl == 0) {
#if GP_ABLATE == 4        // (timing only: no chunk barriers, no DMA behind the first chunk)
        if constexpr (c == 0) { dma_wait(); __syncthreads(); }
#else
        dma_wait();       // this wave's pieces of chunk c have landed
        __syncthreads();  // chunk c visible; everyone finished reading chunk c-1
        if constexpr (c + 1 < NCH)
          stage_chunk<T>(frags + (c + 1) * kChunk * 64, &s_fr[(c + 1) & 1][0], wave, lane);
#endif
        // the chunk's first A operands (the ring below keeps kAhead - 1 reads in flight)
        static_for<kAhead - 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
#if GP_ABLATE >= 3
          if constexpr (j < kChunk && f + j < NF) afr[j % kAhead] = kv[j % NK];
#else
          if constexpr (j < kChunk && f + j < NF) afr[j % kAhead] = s_fr[c & 1][j * 64 + lane];
#endif
        });
      }
      // A operands are read kAhead - 1 fragments ahead of their matrix instruction (inside the
      // chunk: the next chunk becomes readable only behind its barrier), so the LDS latency
      // hides behind the matrix instructions in between instead of in front of each pair
#if GP_ABLATE >= 3
      if constexpr (fl + kAhead - 1 < kChunk && f + kAhead - 1 < NF)
        afr[(fl + kAhead - 1) % kAhead] = kv[(f + 1) % NK];
#else
      if constexpr (fl + kAhead - 1 < kChunk && f + kAhead - 1 < NF)
        afr[(fl + kAhead - 1) % kAhead] = s_fr[c & 1][(fl + kAhead - 1) * 64 + lane];
#endif
      if constexpr (I == J && s == 0) acc = acc_t{T(0), T(0), T(0), T(0)};
      // (k-steps s >= KL of the last training block are padding: not issued)
      if constexpr (4 * I + s < NK) acc = R::mfma(afr[fl % kAhead], kv[4 * I + s], acc);
#if GP_RING_PIN
      // the order written here is the order issued: left alone, the fp32 build pairs two reads into one
      // ds_read2st64_b32 on a fixed register pair and waits for it in front of its two matrix instructions
      // (read, s_waitcnt lgkmcnt(0), mfma, mfma) -- the ring gone, one LDS latency exposed per pair
      if constexpr (sizeof(T) == 4 || GP_RING_PIN > 1) __builtin_amdgcn_sched_barrier(0);
#endif
      if constexpr (I == NB - 1 && s == 3) {
        // rows of the last column block beyond its KL registers are padding too (exactly 0)
#pragma unroll
        for (int r = 0; r < (J == NB - 1 ? KL : 4); ++r) vacc = fma(acc[r], kv[4 * J + r], vacc);
      }
    });
    rbuf ^= 1;
    GP_STAMP(4);   // phase B
    vacc = xor_reduce_groups(vacc);
    if (m < p.M && g == 1) o_var[m] = b - vacc + poison;

    e = e_next;
    grp = grp_next;
    GP_STAMP(5);   // variance reduction and store
  }
#if GP_STAMPS
  if (lane == 0 && p.dbg) {
    for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&p.dbg[k_], seg_sum[k_]);
    atomicAdd(&p.dbg[7], 1ull);   // wave count
  }
#endif
}

}  // namespace gpk
