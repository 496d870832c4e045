// Training objective for n_train <= 256 with the matrix RESIDENT in registers.
//
// likelihood_kernel (gp_train_kernel.hpp) keeps Q in an N x N workspace in global memory and
// makes ceil(N / 8) read-modify-write passes over it; a batch of per-band emulators then streams
// 3.9 TB/s of workspace traffic from the Infinity Cache and that, not arithmetic, is its time
// (17.3 ms for 2101 sets of N = 250).  Here the same elimination -- in-place Gauss-Jordan without
// pivoting, 8 pivots per pass, columns taken from the pivot rows by the signed symmetry of the
// partly inverted matrix (see gp_train_kernel.hpp; the arithmetic is unchanged) -- runs on a
// matrix that never leaves the CU:
//
//   * the symmetric matrix is padded to 256 x 256 (identity outside N) and cut into 16 x 16
//     tiles; only the 136 tiles of the lower triangle are kept, each in the matrix core's C/D
//     register layout (4 doubles per lane).  One workgroup of 8 waves per theta; wave w owns block
//     rows w and 15 - w: 17 tiles = 136 registers per lane, the same for every wave;
//   * a pass stages the 8 pivot rows in LDS (rows from the tiles of their block row, the part
//     right of the diagonal from the COLUMNS of the tiles below it), runs the 8 scalar steps on
//     that panel with likelihood_kernel's arithmetic (one thread per column; each wave carries
//     the 8 x 8 pivot block along in its first lanes and reads every step's pivot row out of it
//     with v_readlane), and applies all 8 rank-1 updates to every tile with two
//     v_mfma_f64_16x16x4_f64:  tile += C^T W,  C[k][i] = -s_i p_k w^(k)_i,  W[k][j] = w^(k)_j,
//     both operands read from the LDS panels (all 38 reads of a wave first, then its 34 matrix
//     instructions back to back, no branch between the tiles); finally the finished pivot rows /
//     columns (a third panel, s_F) replace the corresponding registers and, in the same sweep over
//     the tiles, the NEXT pass's pivot rows are staged (round 3: one test per tile instead of four).
//     Two barriers per pass: the staging writes s_R, which the update phase never reads;
//   * Q itself is built straight into the tiles; invQt = invQ t is summed from the finished tiles in
//     registers (row sums inside the owning wave, column sums through a per-wave LDS strip, added in
//     wave order); the lower triangle of the inverse is written to global memory once (the mirror
//     image too when the caller wants the matrix) and the gradient sums (likelihood_grad_kernel's
//     work) follow in the same kernel from that L2-hot copy: one kernel per evaluation, half the pairs.
//
// Work per theta: 32 passes x 17 tiles x 2 matrix instructions per wave; the per-pass panel steps
// are latency-bound and run on one wave / 256 threads while the others wait at the barrier.
// Running the steps of pass p + 1 beside the matrix instructions of pass p ("look-ahead") was built twice in
// round 3 and measured slower both times (profiles/r03_train_kernel.txt): an fp64 matrix instruction holds a
// SIMD's pipe for 64 cycles and no vector instruction runs beside it, so the steps' dependent chain, sharing its
// SIMD with a wave that issues matrix instructions back to back, advances one instruction per 64 cycles.
#pragma once
#include <hip/hip_runtime.h>
#include "gp_predict_kernel.hpp"
#include "gp_train_args.hpp"
#include "gp_train_kernel.hpp"

namespace gpk {

constexpr int tmNP = 256;          // padded matrix side
constexpr int tmNB = tmNP / 16;    // 16 block rows
constexpr int tmTiles = tmNB + 1;  // tiles per wave (block rows w and 15 - w of the lower triangle)
constexpr int tmLd = 272;          // LDS row stride of the panels, in doubles: rows g and g + 1 of a
                                   // matrix-core operand read land 32 banks apart (conflict-free)
constexpr int tmThreads = 512;
constexpr int tmB = 8;             // pivots per pass (== tkB)

// GP_TRAIN_STAMPS=1: diagnostic build (tools/train_stamps.py).  Wave 0 of workgroup 0 adds the
// shader cycles of each segment to TrainArgs::dbg: [0] set-up + Q build, [1] pivot rows -> LDS,
// [3] panel steps, [4] tile updates, [5] write-out + invQt, [6] gradient + sums.
// Never used for timing or results (cdna_hip_programming.md section 7, in-kernel stamps).
#ifndef GP_TRAIN_STAMPS
#define GP_TRAIN_STAMPS 0
#endif
#if GP_TRAIN_STAMPS
#define TM_STAMP(seg)                                                                    \
  do {                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    unsigned long long now_;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    seg_sum[seg] += now_ - seg_t0;                                                       \
    seg_t0 = now_;                                                                       \
  } while (0)
#else
#define TM_STAMP(seg) do { } while (0)
#endif
// GP_TRAIN_STAMPS=2: the anatomy of a pass instead -- 1 rows -> LDS (nothing with GP_TRAIN_MERGED_SWEEP), 2 the barrier
// in front of the steps, 3 steps + barrier, 4 operand reads of the update, 5 its matrix instructions, 6 pivot rows /
// columns replaced (+ the next pass's rows -> LDS), 7 everything behind the passes; wave GP_TRAIN_STAMP_WAVE of workgroup 0.
#ifndef GP_TRAIN_STAMP_WAVE
#define GP_TRAIN_STAMP_WAVE 0
#endif
#if GP_TRAIN_STAMPS == 2
#define TM_FINE(seg) TM_STAMP(seg)
#define TM_COARSE(seg) do { } while (0)
#else
#define TM_FINE(seg) do { } while (0)
#define TM_COARSE(seg) TM_STAMP(seg)
#endif
#ifndef GP_TRAIN_MERGED_SWEEP
#define GP_TRAIN_MERGED_SWEEP 1  // a pass's finished rows / columns and the next pass's staging in ONE sweep over the tiles
#endif

template <int DM>
__global__ __launch_bounds__(tmThreads, 2) void likelihood_mfma_kernel(TrainArgs p) {
  typedef f64x4 acc_t;
  __shared__ double s_R[tmB * tmLd];      // the panel: the pivot rows as staged
  __shared__ double s_W[tmB * tmLd];      // w^(k): pivot row k / pivot at its own step
  __shared__ double s_C[tmB * tmLd];      // -s_i p_k w^(k)_i: the A operand of the update
  __shared__ double s_F[tmB * tmLd];      // the finished pivot rows
  __shared__ double s_vec[tmNP];          // targets
  __shared__ double s_a[tmNP];            // invQt
  __shared__ double s_piv[tmNP];          // the pivots (their logs sum to logdet Q)
  __shared__ double s_e[tkMaxD + 2];
  __shared__ double s_red[tmThreads / 64];
  __shared__ double s_x[DM * tmNP];       // inputs, transposed [d][i], zero beyond N

  const int N = p.N, D = p.D;
  const int e = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, ml = lane & 15;
  double* A = p.work + (long long)e * N * N;
  const double* th = p.theta + (long long)e * (D + 2);
  const double* tg = p.targets + (long long)e * p.targets_stride;

#if GP_TRAIN_STAMPS
  unsigned long long seg_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long seg_t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seg_t0)::"memory");
#endif
  if (tid < D + 2) s_e[tid] = exp(th[tid]);
  for (int i = tid; i < tmNP; i += tmThreads) s_vec[i] = i < N ? tg[i] : 0.0;
  for (int idx = tid; idx < DM * tmNP; idx += tmThreads) {
    const int d = idx / tmNP, i = idx - d * tmNP;
    s_x[idx] = (d < D && i < N) ? p.inputs[(long long)i * D + d] : 0.0;
  }
  __syncthreads();
  const double b = s_e[D], noise = s_e[D + 1];

  // block row / column of this wave's tile t (wave-uniform)
  auto tile_R = [&](int t) { return t <= w ? w : tmNB - 1 - w; };
  auto tile_C = [&](int t) { return t <= w ? t : t - w - 1; };

  // ---- Q, straight into the tiles ------------------------------------------------------------
  // register r of lane (g, ml) of tile (R, C) is element (16 R + g + 4 r, 16 C + ml)
  double tl[tmTiles][4];
  static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    int wv = w;
    asm volatile("" : "+s"(wv));           // (a tile's coordinates are read for that tile: shared between the tiles of a
    const int R = t <= wv ? wv : tmNB - 1 - wv, C = t <= wv ? t : t - wv - 1;   // block row they stay live, 16 DM registers)
    const int j = 16 * C + ml;
    double xj[DM];
    static_for<DM>([&](auto dc) __attribute__((always_inline)) {
      constexpr int d = decltype(dc)::value;
      xj[d] = s_x[d * tmNP + j];
    });
    static_for<4>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      const int i = 16 * R + g + 4 * r;
      double r2 = 0.0;
      static_for<DM>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        const double dl = s_x[d * tmNP + i] - xj[d];
        r2 = fma((d < D ? s_e[d] : 0.0) * dl, dl, r2);
      });
      const double q = b * exp(-0.5 * r2) + (i == j ? noise : 0.0);
      tl[t][r] = (i < N && j < N) ? q : (i == j ? 1.0 : 0.0);
    });
    // the tile is FINISHED here, before the next one's coordinates are read: left to the scheduler, every tile's LDS
    // reads go first and their results are spilled until the arithmetic gets to them
    asm volatile("" : "+v"(tl[t][0]), "+v"(tl[t][1]), "+v"(tl[t][2]), "+v"(tl[t][3]));
  });

  // ---- in-place Gauss-Jordan inversion, 8 pivots per pass --------------------------------------
  // One pass.  h = which half of the pivots' 16-block they are (its rows 8 h .. 8 h + 7), a
  // compile-time constant: the registers that hold pivot rows (2 h and 2 h + 1 of a tile) must be
  // named statically or the tiles would be addressed, i.e. live in scratch memory.
  TM_STAMP(0);
  // Which of this wave's tiles a pass touches specially (all wave-uniform):
  //   tiles [tlo, thi] lie in the pivots' block row (this wave owns it, else the range is empty);
  //   tiles t1 / t2 are the ones BELOW the pivot block in its block column (rows w / 15 - w);
  //   the diagonal tile of the pivot block is thi;
  // and the same as bit masks over the tile index: one scalar bit test per tile and question.
  struct PassGeo { int Ib, tlo, t1; unsigned rowmask, belowmask, colmask; };
  auto geo_of = [&](const int k0, const int wv) __attribute__((always_inline)) {
    PassGeo q;
    q.Ib = k0 >> 4;                   // block row of the pivots
    const int own = (q.Ib == wv) ? 0 : (q.Ib == tmNB - 1 - wv) ? 1 : -1;
    q.tlo = own == 0 ? 0 : own == 1 ? wv + 1 : 1;
    const int thi = own == 0 ? wv : own == 1 ? tmNB : 0;
    q.t1 = q.Ib < wv ? q.Ib : -1;
    const int t2 = q.Ib < tmNB - 1 - wv ? wv + 1 + q.Ib : -1;
    q.rowmask = own < 0 ? 0u : ((2u << thi) - 1u) & ~((1u << q.tlo) - 1u);
    q.belowmask = (q.t1 >= 0 ? 1u << q.t1 : 0u) | (t2 >= 0 ? 1u << t2 : 0u);
    q.colmask = q.belowmask | (own < 0 ? 0u : 1u << thi);     // + the diagonal tile
    return q;
  };
  const int lrow = g * tmLd + ml;              // this lane's offset inside a panel row pair
  // (1) a tile's share of the 8 pivot rows -> s_R.  Columns up to the pivot block come from the tiles of block
  // row Ib (registers 2 h and 2 h + 1 hold rows 8 h + g and 8 h + 4 + g); columns right of it from
  // the pivot COLUMNS of the tiles below, A[k][j] = A[j][k] while neither is eliminated.
  auto stage_tile = [&](auto tc, auto hc, const PassGeo& q, const int wv) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    constexpr int h = decltype(hc)::value;
    if (q.rowmask & (1u << t)) {
      const int C = t - q.tlo;
      s_R[lrow + 16 * C] = tl[t][2 * h];
      s_R[lrow + 4 * tmLd + 16 * C] = tl[t][2 * h + 1];
    }
    if (q.belowmask & (1u << t)) {
      const int R = t == q.t1 ? wv : tmNB - 1 - wv;
      if ((ml >> 3) == h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s_R[(ml & 7) * tmLd + 16 * R + g + 4 * r] = tl[t][r];
      }
    }
  };
  // (5) a tile's finished pivot rows and columns (from s_F) into its registers
  auto replace_tile = [&](auto tc, auto hc, const PassGeo& q, const int wv, const int k0) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    constexpr int h = decltype(hc)::value;
    const int R = t <= wv ? wv : tmNB - 1 - wv, C = t <= wv ? t : t - wv - 1;
    if (q.rowmask & (1u << t)) {                // pivot rows of this tile
      tl[t][2 * h] = s_F[lrow + 16 * C];
      tl[t][2 * h + 1] = s_F[lrow + 4 * tmLd + 16 * C];
    }
    if (q.colmask & (1u << t)) {                // pivot columns: the signed transpose of the rows
      if ((ml >> 3) == h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int il = g + 4 * r, i = 16 * R + il;
          const bool pivot_row = (R == q.Ib) && ((il >> 3) == h);
          if (!pivot_row) {
            const double rv = s_F[(ml & 7) * tmLd + i];
            tl[t][r] = (i < k0) ? rv : -rv;
          }
        }
      }
    }
  };
  auto pass = [&](const int k0, auto hc) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    typedef std::integral_constant<int, 1 - h> HN;
    // wv is made opaque once per pass: everything derived from it (tile coordinates, LDS
    // addresses) is then recomputed per pass with a few scalar instructions instead of being
    // hoisted out of the pass loop into ~100 registers, which would push tiles into scratch.
    int wv = w;
    asm volatile("" : "+s"(wv));
    const PassGeo q = geo_of(k0, wv);
#if !GP_TRAIN_MERGED_SWEEP
    static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) { stage_tile(tc, hc, q, wv); });
#endif
    TM_FINE(1);
    __syncthreads();
    TM_COARSE(1);
    TM_FINE(2);
    // (2) + (3) the 8 scalar steps on the panel, one thread per column (waves 0..3).  Every wave
    // also carries the 8 x 8 pivot block, column c in lane c, and takes each step's pivot row out
    // of it with v_readlane (scalar registers): no LDS round trip and no barrier between the steps
    // on the block and the steps on the columns.  Step k (pivot p = B[k][k], row r_k = B[k][:]):
    //   f_r = -r_kr (r < k) | r_kr (r > k)   [column k, from the pivot row by the signed symmetry]
    //   w = x_k / p;  x_r -= f_r w (r != k);  x_k = w          for every column x (block and panel)
    //   block column k itself: x_k = 1 / p,  x_r = -f_r / p.
    if (tid < tmNP) {
      const int j = tid;
      double col[tmB], blk[tmB];
#pragma unroll
      for (int r = 0; r < tmB; ++r) {
        col[r] = s_R[r * tmLd + j];
        blk[r] = s_R[r * tmLd + k0 + (lane & 7)];
      }
      const bool inK = (j >= k0 && j < k0 + tmB);
      const double sg = inK ? 0.0 : (j < k0 ? 1.0 : -1.0);   // -s_j; pivot rows take no update
      static_for<tmB>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        double rowk[tmB];                                     // pivot row k of the block: uniform
#pragma unroll
        for (int r = 0; r < tmB; ++r) {
          const unsigned long long u = (unsigned long long)__double_as_longlong(blk[k]);
          const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, r);
          const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), r);
          rowk[r] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
        const double piv = rowk[k];
        double ip = __builtin_amdgcn_rcp(piv);                // v_rcp_f64 + two Newton steps
        ip = fma(fma(-piv, ip, 1.0), ip, ip);
        ip = fma(fma(-piv, ip, 1.0), ip, ip);
        const double wk = col[k] * ip, wb = blk[k] * ip;
        const bool pc = (lane & 7) == k;                      // this lane holds block column k
        s_W[k * tmLd + j] = inK ? 0.0 : wk;                   // pivot columns take no update
        s_C[k * tmLd + j] = sg * piv * wk;
#pragma unroll
        for (int r = 0; r < tmB; ++r) {
          if (r != k) {
            const double f = r < k ? -rowk[r] : rowk[r];
            col[r] = fma(-f, wk, col[r]);
            blk[r] = pc ? -f * ip : fma(-f, wb, blk[r]);
          }
        }
        col[k] = wk;
        blk[k] = pc ? ip : wb;
        if (tid == 0) s_piv[k0 + k] = piv;
      });
      // finished pivot rows: panel columns from their threads, the pivot block from wave 0
#pragma unroll
      for (int r = 0; r < tmB; ++r) {
        if (!inK) s_F[r * tmLd + j] = col[r];
        if (tid < tmB) s_F[r * tmLd + k0 + tid] = blk[r];
      }
    }
    __syncthreads();
    TM_STAMP(3);
    // (4) the update of every tile and the finished pivot rows and columns.  All operand reads
    // first (one LDS latency for the lot), then the matrix instructions back to back.
    // (Tried and not kept, profiles/r03_train_kernel.txt: the finished rows / columns put in BEFORE the matrix
    // instructions by waves 4..7 -- legal, pivot rows have a zero A operand and pivot columns a zero B operand -- and a
    // bare s_barrier between the operand reads and the matrix instructions.  While a 64-cycle fp64 matrix
    // instruction of the other wave holds the SIMD's pipe a wave's vector instructions get in one per 64 cycles, so
    // whatever is moved under the other wave's matrix phase takes as long as that phase.)
    auto replace_pivots = [&]() __attribute__((always_inline)) {
      static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) { replace_tile(tc, hc, q, wv, k0); });
    };
    {
      const int Rw = wv, Rv = tmNB - 1 - wv;
      const double aw0 = s_C[lrow + 16 * Rw], aw1 = s_C[lrow + 4 * tmLd + 16 * Rw];
      const double av0 = s_C[lrow + 16 * Rv], av1 = s_C[lrow + 4 * tmLd + 16 * Rv];
      double bq[tmTiles][2];
      static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        const int C = t <= wv ? t : t - wv - 1;
        bq[t][0] = s_W[lrow + 16 * C];
        bq[t][1] = s_W[lrow + 4 * tmLd + 16 * C];
      });
      TM_FINE(4);
      // every tile, also those wholly in the identity padding (their operands are zero): no branch
      // between the tiles, so the 34 matrix instructions of a wave issue back to back
      static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        const bool first = t <= wv;
        const double a0 = first ? aw0 : av0, a1 = first ? aw1 : av1;
        acc_t acc = {tl[t][0], tl[t][1], tl[t][2], tl[t][3]};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bq[t][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bq[t][1], acc, 0, 0, 0);
        tl[t][0] = acc[0]; tl[t][1] = acc[1]; tl[t][2] = acc[2]; tl[t][3] = acc[3];
      });
    }
#if GP_TRAIN_STAMPS == 2
    asm volatile("" :: "v"(tl[tmTiles - 1][0]));     // (the last matrix instruction's result has arrived)
#endif
    TM_FINE(5);
#if GP_TRAIN_MERGED_SWEEP
    {
      // (5) and the NEXT pass's (1) in one sweep over the tiles: the tiles either touches are nearly the same ones
      // (always, when the next pivots are the other half of the same 16-block), a wave has one to three of them
      // unless it owns the block row, and a tile's tests cost more than its work -- so one test per tile decides
      // whether any of the four questions is asked.
      PassGeo qn = geo_of(k0 + tmB, wv);
      if (k0 + tmB >= N) qn.rowmask = qn.belowmask = 0u;
      const unsigned any = q.rowmask | q.colmask | qn.rowmask | qn.belowmask;
      static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) {
        if (any & (1u << decltype(tc)::value)) {
          replace_tile(tc, hc, q, wv, k0);
          stage_tile(tc, HN{}, qn, wv);
        }
      });
    }
#else
    replace_pivots();
#endif
    TM_FINE(6);
    // no barrier here: the next pass's step (1) writes s_R only, which nobody reads any more, and
    // its barrier separates this pass's readers of s_W / s_C / s_F from the next pass's writers
    TM_COARSE(4);
  };
#if GP_TRAIN_MERGED_SWEEP
  {
    int wv = w;
    asm volatile("" : "+s"(wv));
    const PassGeo q0 = geo_of(0, wv);
    static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) { stage_tile(tc, std::integral_constant<int, 0>{}, q0, wv); });
  }
#endif
  for (int kb = 0; kb < N; kb += 2 * tmB) {
    pass(kb, std::integral_constant<int, 0>{});
    if (kb + tmB < N) pass(kb + tmB, std::integral_constant<int, 1>{});
  }

  __syncthreads();                     // the last pass's readers of the panels are done
  // ---- the inverse -> global memory: the lower triangle in 128-byte runs (the gradient sums below read
  // it back from L2); the mirror image (8-byte stores, a row apart) only when the caller wants the matrix
  const bool full = p.full_inverse != 0;
  static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) {
    constexpr int t = decltype(tc)::value;
    const int R = tile_R(t), C = tile_C(t);
    const int j = 16 * C + ml;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * R + g + 4 * r;
      if (i < N && j < N) {
        if (R != C || j <= i || full) A[(long long)i * N + j] = tl[t][r];
        if (R != C && full) A[(long long)j * N + i] = tl[t][r];
      }
    }
  });

  // ---- invQt = invQ t straight from the register tiles ------------------------------------------------
  // Tile (R, C), lane (g, ml), register r is element (i = 16 R + g + 4 r, j = 16 C + ml).  A tile gives
  // its rows sum_j v_ij t_j and, below the diagonal, its columns sum_i v_ij t_i (the mirror image);
  // diagonal tiles hold both triangles and give rows only.  A block row belongs to one wave, so its row
  // sums are finished inside the wave (accumulated over the tiles per lane, reduced over the 16 lanes of
  // a group at the end); column sums go to a per-wave strip in LDS (s_W) and are added in wave order.
  {
    double* colw = s_W + w * tmNP;
    for (int i = lane; i < tmNP; i += 64) colw[i] = 0.0;
    double rowA[4] = {0.0, 0.0, 0.0, 0.0}, rowB[4] = {0.0, 0.0, 0.0, 0.0};   // block rows w and 15 - w
    static_for<tmTiles>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      int wv = w;
      asm volatile("" : "+s"(wv));                     // (per-tile addresses are not to be hoisted: see pass)
      const int R = t <= wv ? wv : tmNB - 1 - wv, C = t <= wv ? t : t - wv - 1;
      const double tj = s_vec[16 * C + ml];
      double cs = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = tl[t][r];
        if (t <= wv) rowA[r] = fma(v, tj, rowA[r]);
        else rowB[r] = fma(v, tj, rowB[r]);
        cs = fma(v, s_vec[16 * R + g + 4 * r], cs);
      }
      if (R != C) {
        cs = xor_reduce_groups(cs);                    // over the lane groups: the 16 rows of the tile
        if (g == 0) colw[16 * C + ml] += cs;           // (this wave's strip: plain read-modify-write)
      }
      asm volatile("" ::: "memory");                   // one tile at a time: keeps the live set small
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        rowA[r] += __shfl_xor(rowA[r], o, 64);
        rowB[r] += __shfl_xor(rowB[r], o, 64);
      }
    }
    if (ml == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_R[16 * w + g + 4 * r] = rowA[r];
        s_R[16 * (tmNB - 1 - w) + g + 4 * r] = rowB[r];
      }
    }
  }
  __syncthreads();
  if (tid < tmNP) {
    double s = s_R[tid];
#pragma unroll
    for (int k = 0; k < tmThreads / 64; ++k) s += s_W[k * tmNP + tid];
    s_a[tid] = s;
  }
  __syncthreads();
  double tq = 0.0, ld = 0.0;
  for (int i = tid; i < N; i += tmThreads) {
    tq = fma(s_vec[i], s_a[i], tq);
    ld += log(s_piv[i]);
    p.invQt[(long long)e * N + i] = s_a[i];
  }

  TM_COARSE(5);
  // ---- gradient ------------------------------------------------------------------------------------
  // With c_ij = (invQt_i invQt_j - invQ_ij) Z_ij (gp_train_kernel.hpp, gradient_sums):
  //   dcost/dtheta_d = e_d / 4 sum_ij c_ij (x_id - x_jd)^2,  dcost/dtheta_D = -1/2 sum_ij c_ij,
  //   dcost/dtheta_D+1 = 1/2 e_{D+1} (tr invQ - invQt.invQt).
  // Over the lower triangle only (an off-diagonal pair counts twice), invQ read back from the L2-hot
  // copy this workgroup has just written: wave w takes rows w, w + 8, ...; a lane the columns lane,
  // lane + 64, ... of a row.  A rolled loop on purpose: from the register tiles instead, the 136 tile
  // registers + the DM accumulators + a row's and a column's coordinates do not fit (2 000 spilled
  // registers), and loading the next row while the current one is worked on was measured slower.
  double acc[DM];
  static_for<DM>([&](auto dc) __attribute__((always_inline)) { acc[decltype(dc)::value] = 0.0; });
  double sumc = 0.0, tr = 0.0, ss = 0.0;      // (the barriers of the invQt step also cover the write-out above)
  for (int i = w; i < N; i += tmThreads / 64) {
    const double ai = s_a[i];
    for (int j = lane; j <= i; j += 64) {
      const double q = A[(long long)i * N + j];
      double dl2[DM];
      double r2 = 0.0;
      static_for<DM>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        const double dl = s_x[d * tmNP + i] - s_x[d * tmNP + j];
        dl2[d] = dl * dl;
        r2 = fma(d < D ? s_e[d] : 0.0, dl2[d], r2);
      });
      const double c = (i == j ? 1.0 : 2.0) * fma(ai, s_a[j], -q) * (b * exp(-0.5 * r2));
      sumc += c;
      static_for<DM>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        acc[d] = fma(c, dl2[d], acc[d]);
      });
      if (i == j) {
        tr += q;
        ss = fma(ai, ai, ss);
      }
    }
  }
  // block sums: tq, ld, sumc, tr, ss and the DM dimension sums, through s_R (free now)
  constexpr int kSums = 5 + DM;
  double vals[kSums] = {tq, ld, sumc, tr, ss};
  static_for<DM>([&](auto dc) __attribute__((always_inline)) { vals[5 + decltype(dc)::value] = acc[decltype(dc)::value]; });
  static_for<kSums>([&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    double v = vals[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) s_R[k * (tmThreads / 64) + w] = v;
  });
  __syncthreads();
  if (tid < kSums) {
    double v = 0.0;
    for (int k = 0; k < tmThreads / 64; ++k) v += s_R[tid * (tmThreads / 64) + k];
    s_W[tid] = v;
  }
  __syncthreads();
  if (tid == 0) p.cost[e] = 0.5 * s_W[1] + 0.5 * s_W[0] + 0.5 * N * 1.8378770664093453;   // log(2 pi)
  double* g_out = p.grad + (long long)e * (D + 2);
  if (tid < D) g_out[tid] = s_e[tid] * s_W[5 + tid] / 4.0;
  if (tid == 0) {
    g_out[D] = -0.5 * s_W[2];
    g_out[D + 1] = 0.5 * s_W[3] * noise - 0.5 * s_W[4] * noise;
  }
#if GP_TRAIN_STAMPS
  TM_COARSE(6);
  TM_FINE(7);
  if (tid == 64 * GP_TRAIN_STAMP_WAVE && e == 0 && p.dbg)
    for (int k_ = 0; k_ < 8; ++k_) p.dbg[k_] = seg_sum[k_];
#endif
}

}  // namespace gpk
