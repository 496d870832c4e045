// Training objective on the GPU, for E hyper-parameter sets at once:
// what GaussianProcess._prepare_likelihood + loglikelihood + partial_devs compute
// (gp_emulator/GaussianProcess.py:52-125), i.e. for each theta
//
//   Z_ij = b exp(-1/2 sum_d e_d (x_id - x_jd)^2),  Q = Z + e_{D+1} I,  invQ = Q^-1,
//   invQt = invQ t,  cost = 1/2 logdet Q + 1/2 t.invQt + n/2 log 2pi,
//   dcost/dtheta_d    = e_d (invQt^T V_d invQt - sum(invQ o V_d)) / 4,  V_d = Delta_d^2 o Z,
//   dcost/dtheta_D    = 1/2 sum(invQ o Z) - 1/2 invQt^T Z invQt,
//   dcost/dtheta_D+1  = 1/2 e_{D+1} (tr invQ - invQt.invQt).
//
// Two kernels per evaluation -- likelihood_kernel (Q, inverse, invQt, cost), then
// likelihood_grad_kernel -- each with one 1024-thread workgroup per theta (the E sets are the parallel dimension: random restarts
// of learn_hyperparameters, or the per-band emulators of tests/test_perband_emulator.py:22-37).
// Q lives in a per-theta N x N workspace in HBM / Infinity Cache / L2 and is inverted in place
// by Gauss-Jordan elimination, tkB pivots per pass over the matrix (Q is symmetric positive
// definite: no pivoting; the pivots give logdet).  fp64 only.
//
// What bounds it: a batch is bound by the traffic of those passes (2 N^2 x 8 B each, N / tkB of
// them per theta: 256 workgroups x 500 KB live at once is beyond the L2s, so they stream from
// the Infinity Cache); a single theta by one CU's L2 port and the latency of each pass.
#pragma once
#include <hip/hip_runtime.h>
#include "gp_predict_kernel.hpp"
#include "gp_train_args.hpp"

namespace gpk {

__device__ inline double block_sum(double v, double* s_red) {
  // sum over the 1024 threads; result valid in every thread
  const int tid = threadIdx.y * tkSide + threadIdx.x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < tkThreads / 64; ++w) t += s_red[w];
  return t;
}

// Gradient of the cost from this workgroup's (i, j) elements, for n_inputs <= DM.  With
// c_ij = (invQt_i invQt_j - invQ_ij) Z_ij the formulas above become
//   dcost/dtheta_d = e_d / 4 sum_ij c_ij (x_id - x_jd)^2,   dcost/dtheta_D = -1/2 sum_ij c_ij.
template <int DM>
__device__ inline void gradient_sums(const double* A, const double* s_x, const double* s_a,
                                     const double* s_e, int N, int D, double b, double noise,
                                     double* g_out, double* s_red) {
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * tkSide + tx;
  double acc[DM];
#pragma unroll
  for (int d = 0; d < DM; ++d) acc[d] = 0.0;
  double sumc = 0.0, tr = 0.0, ss = 0.0;
  for (int i = ty; i < N; i += tkSide) {
    const double ai = s_a[i];
    for (int j = tx; j < N; j += tkSide) {
      const double q = A[(long long)i * N + j];
      double dl2[DM];
      double r2 = 0.0;
#pragma unroll
      for (int d = 0; d < DM; ++d) {
        const double dl = (d < D) ? s_x[d * N + i] - s_x[d * N + j] : 0.0;
        dl2[d] = dl * dl;
        r2 = fma(d < D ? s_e[d] : 0.0, dl2[d], r2);
      }
      const double c = fma(ai, s_a[j], -q) * (b * exp(-0.5 * r2));
      sumc += c;
#pragma unroll
      for (int d = 0; d < DM; ++d) acc[d] = fma(c, dl2[d], acc[d]);
      if (i == j) {
        tr += q;
        ss = fma(ai, ai, ss);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < DM; ++d) {
    if (d < D) {                           // D is uniform: every thread takes the same branches
      const double a = block_sum(acc[d], s_red);
      if (tid == 0) g_out[d] = s_e[d] * a / 4.0;
    }
  }
  sumc = block_sum(sumc, s_red);
  tr = block_sum(tr, s_red);
  ss = block_sum(ss, s_red);
  if (tid == 0) {
    g_out[D] = -0.5 * sumc;
    g_out[D + 1] = 0.5 * tr * noise - 0.5 * ss * noise;
  }
}

__global__ __launch_bounds__(tkThreads) void likelihood_kernel(TrainArgs p) {
  __shared__ double s_P[tkB][tkB];     // the pivot block after its tkB steps
  __shared__ double s_F[tkB][tkB];     // step k's column factors inside the block
  __shared__ double s_ip[tkB], s_pk[tkB];   // 1 / pivot, pivot of each step
  __shared__ double s_vec[tkMaxN];     // targets
  __shared__ double s_a[tkMaxN];       // invQt
  __shared__ double s_piv[tkMaxN];     // the pivots (their logs sum to logdet Q)
  __shared__ double s_e[tkMaxD + 2];
  __shared__ double s_red[tkThreads / 64];
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // likelihood_lds_bytes()
  const int N = p.N, D = p.D;
  double* s_R = s_dyn;                 // [tkB][N] the panel: pivot rows, updated in place
  double* s_W = s_R + tkB * N;         // [tkB][N] w^(k): pivot row k / pivot at its own step
  double* s_x = s_W + tkB * N;         // [D][N] inputs, transposed

  const int e = blockIdx.x;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * tkSide + tx;
  double* A = p.work + (long long)e * N * N;
  const double* th = p.theta + (long long)e * (D + 2);
  const double* tg = p.targets + (long long)e * p.targets_stride;

  if (tid < D + 2) s_e[tid] = exp(th[tid]);
  for (int i = tid; i < N; i += tkThreads) s_vec[i] = tg[i];
  for (int idx = tid; idx < N * D; idx += tkThreads) {
    const int i = idx / D, d = idx - i * D;
    s_x[d * N + i] = p.inputs[idx];
  }
  __syncthreads();
  const double b = s_e[D], noise = s_e[D + 1];

  // ---- Q ---------------------------------------------------------------------------------
  for (int i = ty; i < N; i += tkSide)
    for (int j = tx; j < N; j += tkSide) {
      double r2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double dl = s_x[d * N + i] - s_x[d * N + j];
        r2 = fma(s_e[d] * dl, dl, r2);
      }
      A[(long long)i * N + j] = b * exp(-0.5 * r2) + (i == j ? noise : 0.0);
    }

  // ---- in-place Gauss-Jordan inversion, tkB pivots per pass over the matrix -----------------
  // Scalar Gauss-Jordan step k (pivot p = A[k][k], row r_k = A[k][:], column c_k = A[:][k]):
  //   A[k][k] <- 1/p,  A[k][j] <- w_j = r_kj / p,  A[i][k] <- -c_ik / p,  A[i][j] <- A[i][j] - c_ik w_j.
  // The partly inverted matrix stays symmetric up to sign -- A[i][j] = -A[j][i] when exactly one
  // of i, j is already eliminated, +A[j][i] otherwise -- so columns are never read: c_ik is taken
  // from the pivot row, c_ik = s_i r_ki with s_i = -1 for eliminated i.  (Using one value for
  // both triangles is also what keeps the elimination backward stable at cond(Q) ~ 1e7.)
  // A pass takes the tkB pivot rows K = k0 .. k0+nbk into LDS, runs those tkB steps on the panel
  // alone -- wave 0 on the nbk x nbk block (which yields every step's pivot and factors f), then
  // one thread per column -- keeping each step's w^(k), and applies all of them to the rest of
  // the matrix at once:  A[i][j] <- A[i][j] - sum_k (s_i p_k w^(k)_i) w^(k)_j.  The arithmetic is
  // that of the scalar algorithm; the matrix is read and written once per tkB pivots.
  for (int k0 = 0; k0 < N; k0 += tkB) {
    const int nbk = (N - k0 < tkB) ? (N - k0) : tkB;
    __syncthreads();                       // the previous pass's updates are complete
    for (int idx = tid; idx < nbk * N; idx += tkThreads) s_R[idx] = A[(long long)k0 * N + idx];
    __syncthreads();
    if (tid < tkB * tkB) {                 // the pivot block: wave 0, one lane per element
      const int r = tid / tkB, c = tid % tkB;
      double v = (r < nbk && c < nbk) ? s_R[r * N + k0 + c] : (r == c ? 1.0 : 0.0);
#pragma unroll 1
      for (int k = 0; k < tkB; ++k) {
        const double piv = __shfl(v, k * tkB + k, 64);
        const double rkc = __shfl(v, k * tkB + c, 64);    // pivot row at column c
        const double rkr = __shfl(v, k * tkB + r, 64);    // pivot row at column r
        const double f = (r < k) ? -rkr : rkr;            // column k at row r, by symmetry
        double ip = __builtin_amdgcn_rcp(piv);            // v_rcp_f64 + two Newton steps
        ip = fma(fma(-piv, ip, 1.0), ip, ip);
        ip = fma(fma(-piv, ip, 1.0), ip, ip);
        const double wc = rkc * ip;
        const double on_row = (c == k) ? ip : wc;
        const double off_row = (c == k) ? -f * ip : fma(-f, wc, v);
        v = (r == k) ? on_row : off_row;
        if (c == 0) s_F[k][r] = (r == k) ? 0.0 : f;
        if (tid == 0) {
          s_ip[k] = ip;
          s_pk[k] = piv;
          if (k < nbk) s_piv[k0 + k] = piv;
        }
      }
      s_P[r][c] = v;
    }
    __syncthreads();
    for (int j = tid; j < N; j += tkThreads) {            // the same steps on panel column j
      double col[tkB];
#pragma unroll
      for (int r = 0; r < tkB; ++r) col[r] = (r < nbk) ? s_R[r * N + j] : 0.0;
      const bool inK = (j >= k0 && j < k0 + nbk);
#pragma unroll
      for (int k = 0; k < tkB; ++k) {
        const double w = col[k] * s_ip[k];
        if (k < nbk) s_W[k * N + j] = w;
#pragma unroll
        for (int r = 0; r < tkB; ++r)
          if (r != k) col[r] = fma(-s_F[k][r], w, col[r]);
        col[k] = w;
      }
#pragma unroll
      for (int r = 0; r < tkB; ++r)
        if (r < nbk) s_R[r * N + j] = inK ? s_P[r][j - k0] : col[r];
    }
    __syncthreads();
    // Everything outside the pivot rows and columns.  Per thread: tkU columns with their w
    // values in registers; the next row's elements are loaded before this row's are used, so
    // loads stay in flight across the loop.
    for (int j0 = tx; j0 < N; j0 += tkSide * tkU) {
      double w[tkU][tkB];
      bool jin[tkU];
#pragma unroll
      for (int u = 0; u < tkU; ++u) {
        const int j = j0 + tkSide * u;
        jin[u] = j < N && !(j >= k0 && j < k0 + nbk);
#pragma unroll
        for (int r = 0; r < tkB; ++r) w[u][r] = (jin[u] && r < nbk) ? s_W[r * N + j] : 0.0;
      }
      double v[tkU], vn[tkU];
#pragma unroll
      for (int u = 0; u < tkU; ++u)
        v[u] = (ty < N && jin[u]) ? A[(long long)ty * N + j0 + tkSide * u] : 0.0;
      for (int i = ty; i < N; i += tkSide) {
        const int in = i + tkSide;
#pragma unroll
        for (int u = 0; u < tkU; ++u)
          vn[u] = (in < N && jin[u]) ? A[(long long)in * N + j0 + tkSide * u] : 0.0;
        const bool iin = !(i >= k0 && i < k0 + nbk);
        const double sg = (i < k0) ? 1.0 : -1.0;          // -s_i: the update subtracts
        double ci[tkB];
#pragma unroll
        for (int r = 0; r < tkB; ++r) ci[r] = (r < nbk) ? sg * s_pk[r] * s_W[r * N + i] : 0.0;
#pragma unroll
        for (int u = 0; u < tkU; ++u) {
          double x = v[u];
#pragma unroll
          for (int r = 0; r < tkB; ++r) x = fma(ci[r], w[u][r], x);
          if (iin && jin[u]) A[(long long)i * N + j0 + tkSide * u] = x;
          v[u] = vn[u];
        }
      }
    }
    // The finished pivot rows, and the pivot columns as their (signed) transpose.
    for (int idx = tid; idx < nbk * N; idx += tkThreads) {
      const int r = idx / N, j = idx - r * N;
      const double rv = s_R[idx];
      A[(long long)k0 * N + idx] = rv;
      if (!(j >= k0 && j < k0 + nbk)) A[(long long)j * N + (k0 + r)] = (j < k0) ? rv : -rv;
    }
  }
  __syncthreads();

  // ---- invQt, cost -----------------------------------------------------------------------
  double tq = 0.0;
  for (int i = tid; i < N; i += tkThreads) {
    double s = 0.0;
    for (int j = 0; j < N; ++j) s = fma(A[(long long)i * N + j], s_vec[j], s);
    s_a[i] = s;
    tq = fma(s_vec[i], s, tq);
  }
  tq = block_sum(tq, s_red);               // (includes the barriers that publish s_a)
  double ld = 0.0;
  for (int i = tid; i < N; i += tkThreads) ld += log(s_piv[i]);
  ld = block_sum(ld, s_red);
  for (int i = tid; i < N; i += tkThreads) p.invQt[(long long)e * N + i] = s_a[i];
  if (tid == 0)
    p.cost[e] = 0.5 * ld + 0.5 * tq + 0.5 * N * 1.8378770664093453;   // log(2 pi)
}

// Second kernel of an evaluation: the gradient, from the finished invQ (p.work) and invQt.
template <int DM>
__global__ __launch_bounds__(tkThreads) void likelihood_grad_kernel(TrainArgs p) {
  __shared__ double s_a[tkMaxN];       // invQt
  __shared__ double s_e[tkMaxD + 2];
  __shared__ double s_red[tkThreads / 64];
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // [D][N] inputs, transposed
  const int N = p.N, D = p.D;
  double* s_x = s_dyn;
  const int e = blockIdx.x;
  const int tid = threadIdx.y * tkSide + threadIdx.x;
  const double* A = p.work + (long long)e * N * N;
  const double* th = p.theta + (long long)e * (D + 2);
  if (tid < D + 2) s_e[tid] = exp(th[tid]);
  for (int i = tid; i < N; i += tkThreads) s_a[i] = p.invQt[(long long)e * N + i];
  for (int idx = tid; idx < N * D; idx += tkThreads) {
    const int i = idx / D, d = idx - i * D;
    s_x[d * N + i] = p.inputs[idx];
  }
  __syncthreads();
  gradient_sums<DM>(A, s_x, s_a, s_e, N, D, s_e[D], s_e[D + 1], p.grad + (long long)e * (D + 2), s_red);
}

}  // namespace gpk
