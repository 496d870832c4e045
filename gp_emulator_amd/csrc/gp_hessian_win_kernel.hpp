// Hessian on the matrix core, WINDOWED form (the default for every matrix-core instance).
//
// hessian_mfma_kernel walks the products block by block: one accumulator at a time runs over ALL
// training points, so the whole weight tile of a wave's 16 rows must sit in registers through the
// matrix phase -- 8 NB registers in fp64 (152 at N = 300), which with t'', G and a training row does
// not fit the 256 registers of a wave at two waves per SIMD.  The wide geometry of that kernel buys
// the registers with occupancy (one wave per SIMD, 512 registers) and pays for it: a lone wave
// reaches 85 % of the matrix pipe and 76 % of the fp64 vector rate and has nothing to cover its own
// LDS and barrier waits with.
//
// Here the ORDER is turned round instead.  The training points are taken in windows of KW k-steps
// (4 KW points per lane group); for each window
//   phase A   the weights of the window only (2 KW registers), two training points in flight;
//   phase B   for every k-step of the window, one matrix instruction into EACH of the NBLK block
//             accumulators (fragments packed k-step-major: (k-step, block)) -- NBLK independent
//             chains, no dependent issue at all.
// All NBLK accumulators stay live (8 NBLK registers: 80 at D = 16), but the weight tile never
// exists as a whole: 80 + 2 KW + t'' + two training rows fit 256 registers, so the kernel runs two
// waves per SIMD without spills.  s and G_d are not accumulated in phase A at all: they come out of
// spare accumulator slots of the diagonal blocks (hess_gslot_*).  The blocks are finished and stored
// after the last window.  Workgroups are four waves (one per SIMD), two per CU (WGeo): the two waves
// of a SIMD belong to different workgroups and drift apart, so one's latency-bound parts (item start,
// finish, stores) run under the other's pipe work.  Measured (profiles/r02_hessian_wide.txt,
// r02_hessian_kernels.txt): config 5 (N = 300, D = 16) 2.25-2.31 ms in fp64 against 2.67-2.9 ms for the
// wide geometry, 1.30 against 1.52-1.76 ms in fp32, and faster or equal at every other compiled shape.
#pragma once
#include "gp_hessian_mfma_kernel.hpp"

#ifndef GP_HESS_WINDOWS
#define GP_HESS_WINDOWS 4
#endif
#ifndef GP_HESS_WIN_GROUP
#define GP_HESS_WIN_GROUP 2
#endif
// Timing-only ablations for tools/ab_bench.py (outputs are wrong when set), a bit mask:
// 1 no chunk barriers in the matrix phase, 2 no stores, 4 no finish at all, 8 fragments staged once per item only
#ifndef GP_HESS_ABL
#define GP_HESS_ABL 0
#endif
// GP_HESS_PIPE = 1: phase A of window q + 1 is issued UNDER the matrix instructions of window q -- one training
// point (k-step) of the next window per k-step of the current one, in the same instruction stream: the fp64
// matrix instructions (ten independent chains, always ready) fill every dependent-issue bubble of the
// vector chain, so a wave keeps the pipe busy by itself instead of relying on its SIMD partner being in a
// complementary phase.  Windows are GP_HESS_PIPE_KW k-steps; two window's weights are live (2 x KW registers
// pairs), one training point in flight.  0: the round-2 order (a window's weights, then its matrix phase).
#ifndef GP_HESS_PIPE
#define GP_HESS_PIPE 1
#endif
#ifndef GP_HESS_PIPE_KW
#define GP_HESS_PIPE_KW 10
#endif
#ifndef GP_HESS_ROW_EARLY
#define GP_HESS_ROW_EARLY 0     // 1: next item's row loaded where the last window begins (A/B: 91 registers spilled)
#endif
#ifndef GP_HESS_ROW_MID
#define GP_HESS_ROW_MID 1       // 1: ... in front of row block GP_HESS_ROW_PASS of the finish; 0: behind the finish
#endif
#ifndef GP_HESS_ROW_PASS
#define GP_HESS_ROW_PASS (NB4 > 1 ? NB4 - 2 : 0)     // (one block earlier: 9 registers spilled at D = 16)
#endif
#ifndef GP_HESS_LDS_OUT
#define GP_HESS_LDS_OUT 1      // 0: the round-2 finish (16-byte stores straight from the accumulators), A/B reference
#endif

namespace gpk {

// G and s ride on the matrix core.  A diagonal 4 x 4 block computes every off-diagonal pair twice --
// accumulator register r of lane group g is (4 bi + r, 4 bi + g), and only r <= g is used -- so its six
// slots with r > g are free.  The windowed kernel's packed products put x''_n (n < D) and 1 (n == D)
// there: slot n sits in diagonal block n / 6 at the (r, g) of hess_gslot_r / _g, and the accumulator comes
// out as G_n = sum_i w_i x''_in, respectively s = sum_i w_i -- the 17 vector instructions per (training
// point, test row) that accumulated them in phase A are gone (6 NB4 >= D + 1 for every compiled D).
__host__ __device__ constexpr int hess_gslot_r(int n) { const int k = n % 6; return k == 0 ? 1 : k < 3 ? 2 : 3; }
__host__ __device__ constexpr int hess_gslot_g(int n) { const int k = n % 6; return k == 0 ? 0 : k < 3 ? k - 1 : k - 3; }
__host__ __device__ constexpr int hess_gslot_block(int n) { return hess_block_index(n / 6, n / 6); }
// the slot that (diagonal block bi, register r, lane group g) carries, or -1
__host__ __device__ constexpr int hess_gslot_of(int bi, int r, int g) {
  if (r <= g) return -1;
  const int k = r == 1 ? 0 : r == 2 ? 1 + g : 3 + g;
  return 6 * bi + k;
}

// fragment (block c, k-step ks = 4 I + s) in the windowed kernel's consumption order
__host__ __device__ constexpr int hess_win_frag_index(int c, int I, int s, int nblk) {
  return (4 * I + s) * nblk + c;
}

// Geometry: FOUR waves per workgroup (one per SIMD) and TWO workgroups per CU.  The two waves of a
// SIMD then belong to different workgroups, whose barriers do not tie them together: while one is
// in its latency-bound parts (item start, reductions, finish and stores: a quarter of an item in
// lockstep) the other keeps the fp64 pipe busy.  80 KB of LDS per workgroup: the training rows at
// their natural stride and two 32-fragment chunks.
struct WGeo {
  static constexpr int kWaves = 4;
  static constexpr int kThreads = kWaves * 64;
  static constexpr int kRowsPerWG = kWaves * kTile;
  static constexpr int kChunk = 32;
};
// workgroups per CU = waves per SIMD (256 registers each; three for fp32 spilled 120 registers)
#ifndef GP_HESS_WG_F32
#define GP_HESS_WG_F32 2
#endif
template <typename T> __host__ __device__ constexpr int win_wg_per_cu() { return sizeof(T) == 4 ? GP_HESS_WG_F32 : 2; }
// Instances that run the windowed kernel: all of them.  It started as the cure for the large fp64
// instances (hess_wide), but with s and G on the matrix core and the decoupled workgroups it beats the
// block-major kernel at every compiled shape in both precisions (profiles/r02_hessian_kernels.txt;
// N = 120, D = 8: 0.49 vs 0.59 ms fp64).  hessian_mfma_kernel stays as the A/B reference (GP_HESS_WIN=0).
template <typename T> __host__ __device__ constexpr bool hess_win(int, int) { return true; }

// LDS row stride of a training point [x'', alpha, h]: D + 2 reals rounded up to 16 bytes
template <typename T> __host__ __device__ constexpr int win_row_stride(int D) {
  return (D + 2 + (16 / (int)sizeof(T)) - 1) / (16 / (int)sizeof(T)) * (16 / (int)sizeof(T));
}

// LDSOUT: the finish assembles the output in LDS and stores whole lines (see the finish); the host picks that
// instance when the caller's rows are exactly D long (d_actual == D) and 16-byte pieces (win_lds_out) and the
// matrix is 16-byte aligned, and runs it on the whole 64-row groups of the call (M a multiple of 64: no row
// guards anywhere in it); the rows beyond the last whole group, and every other call, go to the instance with
// the direct stores of round 2.
template <typename T> __host__ __device__ constexpr bool win_lds_out(int D) {
  return GP_HESS_LDS_OUT && D % (16 / (int)sizeof(T)) == 0;
}
// KL: live k-steps of the last 16-block (1..4).  The kernel is compiled per block count NB; a training set that
// leaves the last k-step(s) of its last block empty (N = 300: 75 of 76, N = 250: 63 of 64) runs the instance that
// issues neither their weights nor their ten matrix instructions (instantiated for the BASELINE shapes only:
// hess_win_short_last).
template <typename T> __host__ __device__ constexpr bool hess_win_short_last(int NB) { return NB == 16 || NB == 19; }
template <typename T, int D, int NB, bool LDSOUT, int KL = 4>
__global__ __launch_bounds__(WGeo::kThreads, (win_wg_per_cu<T>()))
void hessian_win_kernel(HessMfmaArgs<T> p) {
  typedef Real<T> R;
  typedef typename R::acc_t acc_t;
  constexpr int kThreads = WGeo::kThreads;
  constexpr int kWaves = WGeo::kWaves;
  constexpr int kRowsPerWG = WGeo::kRowsPerWG;
  constexpr int NP = 16 * NB;
  constexpr int DS = win_row_stride<T>(D);          // LDS image
  constexpr int DSG = row_stride(D);                // packed global image
  constexpr int NB4 = hess_nb4(D);
  constexpr int NBLK = hess_blocks(D);
  constexpr int NKS = 4 * (NB - 1) + KL;                        // k-steps
  constexpr bool kPipe = GP_HESS_PIPE != 0;
  constexpr int NW = kPipe ? (NKS + GP_HESS_PIPE_KW - 1) / GP_HESS_PIPE_KW : (GP_HESS_WINDOWS < NKS ? GP_HESS_WINDOWS : NKS);
  constexpr int KW = (NKS + NW - 1) / NW;                       // k-steps per window (the last may be short)
  constexpr int NF = NKS * NBLK;
  // kPipe, LDSOUT: the k-step at which the next item's test row is loaded: the first of the last window, but behind
  // the chunk barrier that publishes the next item's number (chunk 1)
  constexpr int kRowLoadKs0 = (NKS - 1) / KW * KW;
  constexpr int kRowLoadKs = kRowLoadKs0 * NBLK > 2 * WGeo::kChunk ? kRowLoadKs0 : (NKS - 1);
  constexpr int kChunk = WGeo::kChunk;
  constexpr int NCH = (NF + kChunk - 1) / kChunk;
  // The finish assembles the wave's output in LDS and writes it as whole lines (below) when the rows of the
  // matrix are whole 16-byte pieces; the other instances keep the direct stores.
  constexpr int VW = 16 / (int)sizeof(T);                       // elements per 16 bytes
  constexpr bool kLdsOut = LDSOUT;
  static_assert(!LDSOUT || win_lds_out<T>(D), "whole-line stores need matrix rows of whole 16-byte pieces");
  constexpr int kStageElems = 2 * kChunk * 64 / kWaves;         // a wave's share of the two fragment buffers
  constexpr int kStagePieces = kChunk * 64 * (int)sizeof(T) / 1024 / (kWaves / 2);   // 1 KiB DMA pieces of chunk 0 in it

  __shared__ __attribute__((aligned(16))) T s_xa[NP * DS];
  __shared__ __attribute__((aligned(16))) T s_fr[2][kChunk * 64];
  __shared__ T s_sd[2 * D + 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ml = lane & 15;
  const int g = lane >> 4;

  for (int i = tid; i < NP * DS; i += kThreads) s_xa[i] = p.xa[(i / DS) * DSG + i % DS];
  if (tid < 2 * D + 1) s_sd[tid] = (tid == 2 * D || (tid % D) < p.d_actual) ? p.sd[tid] : T(0);
  __syncthreads();
  const T b = s_sd[2 * D];

  const long long n_groups = (p.M + kRowsPerWG - 1) / kRowsPerWG;
#if GP_STAMPS   // diagnostic build (tools/hess_stamps.py): [0] item start, [1] phase A, [2] phase B, [3] reductions, [4] finish + stores
  unsigned long long seg_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long seg_t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg_t0)::"memory");
  const unsigned long long wave_t0 = seg_t0;
  const unsigned long long real_t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, one counter for the chip
#endif
  // the lane's raw test row, loaded one item ahead: at the end of the last window's phase A, when the
  // registers of the training row are free again, so that the HBM latency of these loads is not the
  // first thing an item waits for
  T rraw[D];
  auto load_row = [&](long long grp_) __attribute__((always_inline)) {
    const long long m_ = grp_ * kRowsPerWG + wave * kTile + ml;
    const long long mc_ = m_ < p.M ? m_ : p.M - 1;
    if constexpr (kLdsOut) {    // d_actual == D, whole groups: every row exists; 16-byte loads
      const T* row = p.testing + m_ * D;
#pragma unroll
      for (int d = 0; d < D; ++d) rraw[d] = row[d];
    } else {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int dc = d < p.d_actual ? d : p.d_actual - 1;     // padded dims: sd = centre = 0
        rraw[d] = p.testing[mc_ * p.d_actual + dc];
      }
    }
  };
  if ((long long)blockIdx.x < n_groups) load_row(blockIdx.x);
  // Items (groups of 64 test rows) are DRAWN, not dealt: the two workgroups of a CU do not advance at the same
  // rate -- the SIMD's arbiter favours the older wave, and with the static round-robin of round 2 the
  // workgroups dispatched first were through their share at 74 % of the launch while the others carried on
  // with a SIMD each to themselves (wave lifetimes 4.0 M vs 5.1 M cycles, profiles/r03_hessian_c5.txt).  A
  // workgroup takes its first item by its index; every further one is gridDim.x + a ticket: one lane draws it
  // (an atomic add) at the top of the item loop, for the NEXT item; wave 0 hands it to the others through LDS
  // in front of the item's second chunk barrier -- a store to a wave-uniform address, no control flow inside
  // the unrolled item -- and every wave reads it in the finish.  Every workgroup draws exactly one ticket
  // beyond the end, so a launch draws n_groups tickets, and the lane that received the last one puts the
  // counter back to 0 for the next launch that uses it.  (p.tickets == nullptr: the same flow with the
  // round-robin numbers.)
  __shared__ unsigned s_next[1 + kWaves];       // [0] the ticket; [1 + wave] where the other waves' copies go
  const bool dyn = p.tickets != nullptr;
  const unsigned last_ticket = (unsigned)(n_groups - 1);
  unsigned drawn = 0;
  unsigned pending = ~0u;
  long long grp = blockIdx.x;
  unsigned items_done = 0;
#if GP_STAMPS == 2
  unsigned long long life_ta = 0, life_tb = 0, life_tc = 0;
#endif
  while (grp < n_groups) {
    ++items_done;
    if (tid == 0) {
      if (dyn && pending == last_ticket) *(volatile unsigned*)p.tickets = 0u;
      const unsigned stat = blockIdx.x + drawn * gridDim.x;
      ++drawn;
      pending = dyn ? atomicAdd(p.tickets, 1u) : stat;
    }
    auto next_grp = [&]() __attribute__((always_inline)) -> long long { return (long long)gridDim.x + s_next[0]; };
    // (opaque per item: the pieces' source addresses are then formed with two scalar adds each where they are
    // used, instead of being hoisted out of the item loop into ~190 scalar registers that live in a vector
    // register's lanes and come back through v_readlane)
    const T* pfr = p.pfrags;
    asm volatile("" : "+s"(pfr));
    if constexpr (kLdsOut) {
      // chunk 0 goes to buffer 0, which is also the finish's staging space of waves 0 and 1 (below): each
      // of the two stages the half of the chunk that lands in its OWN region, as soon as it has left its
      // own finish -- no barrier here; waves 2 and 3 join at the chunk's barrier
      if (wave < kWaves / 2)
        stage_pieces<T, kStagePieces>(pfr, &s_fr[0][0], wave * kStagePieces, lane);
    } else {
      // chunk 0 goes to buffer 0.  With an even number of chunks per item the previous item's last chunk
      // sat in buffer 1, and buffer 0's last readers (the chunk before it) passed a barrier since: no
      // barrier here, the waves of the workgroup may run into the next item a finish apart
      if constexpr (NCH % 2 == 1) __syncthreads();
      stage_chunk<T, kWaves, kChunk>(pfr, &s_fr[0][0], wave, lane);
    }

    T t[D];
    T gm = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      t[d] = s_sd[d] * (rraw[d] - s_sd[D + d]);
      gm = fma(t[d], t[d], gm);
    }
    gm *= T(-0.5);
    T kvw[kPipe ? 2 : 1][KW];   // the weights of the current window (kPipe: and of the next one)
    acc_t accs[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c) accs[c] = acc_t{T(0), T(0), T(0), T(0)};

    // phase A of window q: its KW k-steps, one training point at a time (two waves per SIMD cover
    // the latencies; there is no register for a second point in flight)
    auto window_weights = [&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      constexpr int GP = GP_HESS_WIN_GROUP;        // training points in flight (their serial chains interleave)
      static_for<(KW + GP - 1) / GP>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j0 = decltype(jc)::value * GP;
        T x[GP][D];
        T al[GP], k[GP];
#pragma unroll
        for (int u = 0; u < GP; ++u) {
          const int ks = q * KW + j0 + u < NKS ? q * KW + j0 + u : NKS - 1;     // (a short tail repeats the last k-step)
          const int i = own_index<T>(ks >> 2, ks & 3, g);
          const T* row = &s_xa[i * DS];
#pragma unroll
          for (int d = 0; d < D; ++d) x[u][d] = row[d];
          al[u] = row[D];
          if constexpr (R::kExpand) k[u] = row[D + 1] + gm;
        }
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
#pragma unroll
        for (int u = 0; u < GP; ++u) {
          if constexpr (!R::kExpand) k[u] *= b;
          // (s and G come out of the matrix phase: hess_gslot_*)
          if (j0 + u < KW && q * KW + j0 + u < NKS) kvw[kPipe ? q & 1 : 0][j0 + u] = k[u] * al[u];
        }
      });
    };

    // kPipe: the weight of ONE k-step (the lane's training point of it), into the buffer of its window
    auto point_weight = [&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      const int i = own_index<T>(ks >> 2, ks & 3, g);
      const T* row = &s_xa[i * DS];
      T x[D];
#pragma unroll
      for (int d = 0; d < D; ++d) x[d] = row[d];
      const T al = row[D];
      T k[1];
      if constexpr (R::kExpand) {
        k[0] = row[D + 1] + gm;
#pragma unroll
        for (int d = 0; d < D; ++d) k[0] = fma(x[d], t[d], k[0]);
      } else {
        T r2 = T(0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const T dl = x[d] - t[d];
          r2 = fma(dl, dl, r2);
        }
        k[0] = T(-0.5) * r2;
      }
      R::template exp_n<1>(k);
      if constexpr (!R::kExpand) k[0] *= b;
      kvw[(ks / KW) & 1][ks % KW] = k[0] * al;
    };

    constexpr int kAhead = GP_AHEAD;
    T afr[kAhead];
    GP_STAMP(0);
    window_weights(std::integral_constant<int, 0>{});
    GP_STAMP(1);
    static_for<NF>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      constexpr int ch = f / kChunk, fl = f % kChunk;
      constexpr int ks = f / NBLK, c = f % NBLK;
      constexpr int q = ks / KW;
      if constexpr (kPipe) {
        // the next window's k-step in the same position, under this k-step's matrix instructions
        if constexpr (c == 0 && ks + KW < NKS) point_weight(std::integral_constant<int, ks + KW>{});
        // the next item's test row: loaded where the last window begins -- no weights are formed any more, so
        // the registers are there, and the window's matrix instructions cover most of the HBM latency
        if constexpr (GP_HESS_ROW_EARLY && kLdsOut && c == 0 && ks == kRowLoadKs) {
          __builtin_amdgcn_sched_barrier(0);
          const long long nx = next_grp();
          load_row(nx < n_groups ? nx : grp);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (!kPipe && c == 0 && ks % KW == 0 && q > 0) {
        GP_STAMP(2);
        window_weights(std::integral_constant<int, q>{});
        // (kLdsOut: the row is loaded in the finish, once t and G are dead -- carried through the finish it
        // went to scratch behind a full wait)
        if constexpr (q == (NKS - 1) / KW && !kLdsOut) {
          const long long nx = next_grp();
          load_row(nx < n_groups ? nx : grp);
        }
        GP_STAMP(1);
      }
      if constexpr (fl == 0) {
        GP_STAMP(2);
        dma_wait();       // this wave's pieces of chunk ch have landed
        if constexpr (ch == (NCH > 1 ? 1 : 0))   // the next item's ticket (returned by now), published by the barrier below
          s_next[wave == 0 ? 0 : 1 + wave] = (unsigned)__builtin_amdgcn_readfirstlane((int)pending);
        if constexpr (!(GP_HESS_ABL & 1)) __syncthreads();  // chunk ch visible; everyone finished reading chunk ch-1
#if GP_STAMPS == 2
        if constexpr (f == 0) life_ta = __builtin_amdgcn_s_memrealtime();
#endif
        GP_STAMP(5);
        if constexpr (ch + 1 < NCH && !(GP_HESS_ABL & 8))
          stage_chunk<T, kWaves, kChunk>(pfr + (ch + 1) * kChunk * 64, &s_fr[(ch + 1) & 1][0], wave, lane);
        static_for<kAhead - 1>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (j < kChunk && f + j < NF) afr[j % kAhead] = s_fr[ch & 1][j * 64 + lane];
        });
      }
      if constexpr (fl + kAhead - 1 < kChunk && f + kAhead - 1 < NF)
        afr[(fl + kAhead - 1) % kAhead] = s_fr[ch & 1][(fl + kAhead - 1) * 64 + lane];
      accs[c] = R::mfma(afr[fl % kAhead], kvw[kPipe ? q & 1 : 0][ks - q * KW], accs[c]);
    });
#if GP_HESS_ABL & 4
    {
      T sum_ = gm;
#pragma unroll
      for (int c = 0; c < NBLK; ++c) sum_ += accs[c][0] + accs[c][1] + accs[c][2] + accs[c][3];
      if (sum_ == T(-12345.678)) p.hess[0] = sum_;
      grp = next_grp();
      continue;
    }
#endif

    GP_STAMP(2);
#if GP_STAMPS == 2
    life_tb = __builtin_amdgcn_s_memrealtime();
#endif
    // (everything only the finish needs is formed here, not carried through the windows: the
    // registers are full there, and what does not fit goes to scratch and comes back slowly)
    const T poison = gm - gm;   // NaN for rows holding a NaN or an infinity (exp_ clamps)
    // s and G_d from their accumulator slots: the value sits in one lane group; zero elsewhere and the
    // sum over the four groups (exact) hands it to all of them
    auto slot_value = [&](auto nc) __attribute__((always_inline)) {
      constexpr int n = decltype(nc)::value;
      const T v = accs[hess_gslot_block(n)][hess_gslot_r(n)];
      return xor_reduce_groups(g == hess_gslot_g(n) ? v : T(0));
    };
    const T mu = slot_value(std::integral_constant<int, D>{}) + poison;
    T ga[D];
    static_for<D>([&](auto dc) __attribute__((always_inline)) { ga[decltype(dc)::value] = slot_value(dc); });
    // the epilogue needs t''_d, G_d for a compile-time d -- the lane's own registers, live through every
    // window anyway -- and for d2 = 4 bj + g, picked out per lane group here
    T tq[NB4], gq[NB4];
#pragma unroll
    for (int j = 0; j < NB4; ++j) tq[j] = gq[j] = T(0);
#pragma unroll
    for (int d = 0; d < D; ++d)
      if ((d & 3) == g) {
        tq[d >> 2] = t[d];
        gq[d >> 2] = ga[d];
      }
    GP_STAMP(3);
    // ---------------- finish and store every block ----------------------------------------
    T sdq[NB4];                 // sqrt(e) of this lane group's column in every block: d2 = 4 bj + g
#pragma unroll
    for (int j = 0; j < NB4; ++j) sdq[j] = (4 * j + g < D) ? s_sd[4 * j + g] : T(0);
    // the row length is made opaque once per item: the store offsets of the 10 blocks are then
    // recomputed here with a few integer instructions instead of being hoisted out of the item loop
    // into two dozen registers that the windows would push to scratch
    int da = p.d_actual;
    asm volatile("" : "+s"(da));
    // block cbv from its accumulator: acc[r] = S2[d][d2] for d = 4 bi + r, d2 = 4 bj + g, test row ml
    auto block_values = [&](auto cbc, T (&v)[4]) __attribute__((always_inline)) {
      constexpr int cbv = decltype(cbc)::value;
      constexpr int bi = hess_block_bi(cbv), bj = hess_block_bj(cbv);
      const acc_t a = accs[cbv];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 4 * bi + r;
        v[r] = T(0);
        if (d < D) {            // (compile time after unrolling)
          const T td = t[d], gd = ga[d];
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
    };
    if constexpr (kLdsOut) {
      static_for<NBLK>([&](auto cbc) __attribute__((always_inline)) {      // finish in place
        T v[4];
        block_values(cbc, v);
        accs[decltype(cbc)::value] = acc_t{v[0], v[1], v[2], v[3]};
      });
      // every wave has read its last fragment: both fragment buffers are free until chunk 0 of the next
      // item is staged (by the waves whose staging space it lands in, after their own finish)
      GP_STAMP(3);
      __syncthreads();
      GP_STAMP(6);
      // Whole-line stores.  Lane (ml, g) holds, over the blocks, complete rows 4 p + g (p = 0 .. NB4 - 1) of
      // test row ml's matrix: left of the diagonal block as the mirror images of blocks (b, p), b < p, from
      // it on as the transposed copies of blocks (p, b).  Row block p of all 16 test rows -- 16 runs of
      // 4 D elements, each contiguous in memory -- is assembled in the wave's share of the fragment
      // buffers (16-byte pieces, XOR-swizzled by test row so that neither side has bank conflicts), read
      // back in memory order and stored 16 bytes per lane: every store instruction writes whole 128-byte
      // lines (1 KiB contiguous per half wave at D = 16) instead of 64 pieces of 64 different lines.
      typedef T vec_t __attribute__((ext_vector_type(VW)));
      constexpr int RB16 = D / VW;                     // pieces per matrix row
      constexpr int PRF = 4 * RB16;                    // pieces per full run
      constexpr bool kXor = PRF % 8 == 0;
      constexpr int RS = kXor ? PRF : PRF + 1;         // run stride in pieces (odd when not swizzled)
      static_assert(kTile * RS * VW <= kStageElems, "a row block of the wave's tile must fit its staging space");
      vec_t* stg = reinterpret_cast<vec_t*>(&s_fr[0][0] + wave * kStageElems);
      int lo = lane;                                   // (opaque per item: the piece addresses below are recomputed
      asm volatile("" : "+v"(lo));                     //  from it with a few integer instructions, not hoisted and spilled)
      const int mlo = lo & 15, glo = lo >> 4;
      const long long m0 = grp * kRowsPerWG + wave * kTile;
      T* out0 = p.hess + m0 * (long long)(D * D);
      static_for<NB4>([&](auto ppc) __attribute__((always_inline)) {
        constexpr int pp = decltype(ppc)::value;
        if constexpr (GP_HESS_ROW_MID && pp == GP_HESS_ROW_PASS) {
          // the next item's test row: most accumulators are dead by now, and the remaining row blocks' LDS round
          // trips and stores cover a part of the HBM latency (loaded earlier it lived in scratch: 91 registers)
          __builtin_amdgcn_sched_barrier(0);
          const long long nx = next_grp();
          load_row(nx < n_groups ? nx : grp);
          __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int nrows = D - 4 * pp < 4 ? D - 4 * pp : 4;
        constexpr int PRp = nrows * RB16;              // pieces per run in this row block
        static_for<NB4>([&](auto bc) __attribute__((always_inline)) {
          constexpr int b = decltype(bc)::value;
          T w[4];
          if constexpr (b < pp) {
            const acc_t a = accs[hess_block_index(b, pp)];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = a[r];
          } else {
            const acc_t a = accs[hess_block_index(pp, b)];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = a[r];
            transpose_groups4(w);                      // (cross-lane: executed by every lane)
            if constexpr (b == pp) {
#pragma unroll
              for (int r = 0; r < 4; ++r) w[r] = (r <= g) ? a[r] : w[r];
            }
          }
          if (g < nrows) {
#pragma unroll
            for (int j = 0; j < 4 / VW; ++j) {
              if (4 * b + VW * j < D) {
                const int pi = glo * RB16 + (4 * b) / VW + j;
                vec_t x;
#pragma unroll
                for (int e = 0; e < VW; ++e) x[e] = w[VW * j + e];
                stg[mlo * RS + (kXor ? (pi ^ (mlo & 7)) : pi)] = x;
              }
            }
          }
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        constexpr int NCk = kTile * PRp;               // 16-byte pieces of this row block
        constexpr int NI = (NCk + 63) / 64;
        vec_t rd[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
          const int c = k * 64 + lo;
          const int row = c / PRp, within = c % PRp;
          if (NCk % 64 == 0 || c < NCk) rd[k] = stg[row * RS + (kXor ? (within ^ (row & 7)) : within)];
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
          const int c = k * 64 + lo;
          const int row = c / PRp, within = c % PRp;
#if !(GP_HESS_ABL & 2)
          if (NCk % 64 == 0 || c < NCk)      // (no row guards: this instance runs whole 64-row groups only)
            *reinterpret_cast<vec_t*>(out0 + (long long)row * (D * D) + 4 * pp * D + within * VW) = rd[k];
#else
          if (rd[k][0] == T(-12345.678)) out0[0] = rd[k][0];
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      });
      if constexpr ((!kPipe || !GP_HESS_ROW_EARLY) && !GP_HESS_ROW_MID) {
        __builtin_amdgcn_sched_barrier(0);
        const long long nx = next_grp();
        load_row(nx < n_groups ? nx : grp);     // next item's test row
      }
    } else {
    const long long m = grp * kRowsPerWG + wave * kTile + ml;
    const bool row_ok = m < p.M;
    T* out = p.hess + (row_ok ? m : p.M - 1) * (long long)da * da;
    const bool vec_ok = (da % (16 / (int)sizeof(T)) == 0) && (((unsigned long long)p.hess & 15) == 0);
    static_for<NBLK>([&](auto cbc) {
      constexpr int cbv = decltype(cbc)::value;
      constexpr int bi = hess_block_bi(cbv), bj = hess_block_bj(cbv);
      const int d2 = 4 * bj + g;
      T v[4];
      block_values(cbc, v);
#if GP_HESS_ABL & 2
      if (v[0] + v[1] + v[2] + v[3] == T(-12345.678)) out[0] = v[0];
      if (vec_ok && v[0] == T(-12345.678) && v[1] == T(-1.5))
#else
      if (vec_ok)
#endif
      {
        // 16-byte stores: the mirror image as it is, the block's own rows after a 4 x 4 transpose
        // across the lane groups (see hessian_mfma_kernel)
        T vt[4] = {v[0], v[1], v[2], v[3]};
        transpose_groups4(vt);
        if constexpr (bi != bj) {
          if (row_ok && d2 < da) store_row4<T>(out + d2 * da + 4 * bi, v, da - 4 * bi);
          if (row_ok && 4 * bi + g < da)
            store_row4<T>(out + (4 * bi + g) * da + 4 * bj, vt, da - 4 * bj);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) vt[r] = (r <= g) ? v[r] : vt[r];
          if (row_ok && d2 < da) store_row4<T>(out + d2 * da + 4 * bi, vt, da - 4 * bi);
        }
      } else if (!(GP_HESS_ABL & 2)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 4 * bi + r;
          if (row_ok && d <= d2 && d2 < da) {
            out[d * da + d2] = v[r];
            if (d != d2) out[d2 * da + d] = v[r];
          }
        }
      }
    });
    }
    GP_STAMP(4);
#if GP_STAMPS == 2
    life_tc = __builtin_amdgcn_s_memrealtime();
#endif
    grp = next_grp();
  }
#if GP_STAMPS == 2
  if (tid == 0 && p.dbg && items_done > 0 && items_done <= 40) {
    unsigned long long* rec = p.dbg + 32 + 4 * 512 + 240 * blockIdx.x + 6 * items_done;
    rec[-4] = life_ta; rec[-3] = life_tb; rec[-2] = life_tc;
  }
#endif
  if (dyn && tid == 0 && pending == last_ticket) *(volatile unsigned*)p.tickets = 0u;
#if GP_STAMPS
#if GP_STAMPS == 2      // (no atomics here: 2 048 waves adding to the same few words as they leave slow down the workgroups
                        //  still running by 2-5 x -- an artefact that looked like a property of the launch's end)
  if (tid == 0 && p.dbg) {
    p.dbg[32 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    p.dbg[33 + 4 * blockIdx.x] = items_done;
    p.dbg[34 + 4 * blockIdx.x] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID, bits 3:0
    p.dbg[35 + 4 * blockIdx.x] = real_t0;
  }
#else
  if (lane == 0 && p.dbg) {
    for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&p.dbg[k_], seg_sum[k_]);
    atomicAdd(&p.dbg[7], 1ull);   // wave count
    // [8] shortest, [9] longest, [10] summed wave lifetime; [11] / [12] summed lifetime and count of the waves of the
    // first gridDim / 2 workgroups (the ones dispatched first: the older wave of each SIMD)
    atomicMin(&p.dbg[8], seg_t0 - wave_t0);
    atomicMax(&p.dbg[9], seg_t0 - wave_t0);
    atomicAdd(&p.dbg[10], seg_t0 - wave_t0);
    if (blockIdx.x < gridDim.x / 2) {
      atomicAdd(&p.dbg[11], seg_t0 - wave_t0);
      atomicAdd(&p.dbg[12], 1ull);
    }
    if (tid == 0) atomicAdd(&p.dbg[blockIdx.x < gridDim.x / 2 ? 13 : 14], (unsigned long long)drawn);   // items of the two halves of the grid
    // [16] earliest / [17] latest wave start, [18] earliest / [19] latest wave end on the chip-wide 100 MHz counter
    const unsigned long long real_t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {       // per workgroup: [32 + 4 b] end, items, XCC id, start
      p.dbg[32 + 4 * blockIdx.x] = real_t1;
      p.dbg[33 + 4 * blockIdx.x] = items_done;
      p.dbg[34 + 4 * blockIdx.x] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID, bits 3:0
      p.dbg[35 + 4 * blockIdx.x] = real_t0;
    }
    atomicMin(&p.dbg[16], real_t0);
    atomicMax(&p.dbg[17], real_t0);
    atomicMin(&p.dbg[18], real_t1);
    atomicMax(&p.dbg[19], real_t1);
  }
#endif
#endif
}

}  // namespace gpk
