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
// One 1024-thread workgroup per theta (the E sets are the parallel dimension: random restarts
// of learn_hyperparameters, or the per-band emulators of tests/test_perband_emulator.py:22-37);
// Q lives in a per-theta N x N workspace in HBM/L2 and is inverted in place by Gauss-Jordan
// elimination (Q is symmetric positive definite: no pivoting; the pivots give logdet).  fp64
// only.  A correctness-first kernel: 2 barriers and one pass over the matrix per pivot.
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

__global__ __launch_bounds__(tkThreads) void likelihood_kernel(TrainArgs p) {
  __shared__ double s_row[tkMaxN];     // pivot row (raw)
  __shared__ double s_col[tkMaxN];     // column factors
  __shared__ double s_vec[tkMaxN];     // targets, later invQt
  __shared__ double s_e[tkMaxD + 2];
  __shared__ double s_red[tkThreads / 64];
  __shared__ double s_logdet;

  const int e = blockIdx.x;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * tkSide + tx;
  const int N = p.N, D = p.D;
  double* A = p.work + (long long)e * N * N;
  const double* th = p.theta + (long long)e * (D + 2);
  const double* tg = p.targets + (long long)e * p.targets_stride;

  if (tid < D + 2) s_e[tid] = exp(th[tid]);
  if (tid == 0) s_logdet = 0.0;
  for (int i = tid; i < N; i += tkThreads) s_vec[i] = tg[i];
  __syncthreads();
  const double b = s_e[D], noise = s_e[D + 1];

  // ---- Q ---------------------------------------------------------------------------------
  for (int i = ty; i < N; i += tkSide)
    for (int j = tx; j < N; j += tkSide) {
      double r2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double dl = p.inputs[i * D + d] - p.inputs[j * D + d];
        r2 = fma(s_e[d] * dl, dl, r2);
      }
      A[(long long)i * N + j] = b * exp(-0.5 * r2) + (i == j ? noise : 0.0);
    }

  // ---- in-place Gauss-Jordan inversion ----------------------------------------------------
  for (int k = 0; k < N; ++k) {
    __syncthreads();                       // the previous step's updates are complete
    for (int j = tid; j < N; j += tkThreads) {
      s_row[j] = A[(long long)k * N + j];
      s_col[j] = A[(long long)j * N + k];
    }
    __syncthreads();
    const double piv = s_row[k];
    const double ip = 1.0 / piv;
    if (tid == 0) s_logdet += log(piv);
    for (int i = ty; i < N; i += tkSide) {
      const double f = s_col[i];
      for (int j = tx; j < N; j += tkSide) {
        const double rk = s_row[j] * ip;
        double v;
        if (i == k) v = (j == k) ? ip : rk;
        else if (j == k) v = -f * ip;
        else v = fma(-f, rk, A[(long long)i * N + j]);
        A[(long long)i * N + j] = v;
      }
    }
  }
  __syncthreads();

  // ---- invQt, cost -----------------------------------------------------------------------
  double tq = 0.0;
  for (int i = tid; i < N; i += tkThreads) {
    double s = 0.0;
    for (int j = 0; j < N; ++j) s = fma(A[(long long)i * N + j], s_vec[j], s);
    s_row[i] = s;                          // invQt (s_row is free now)
    tq = fma(s_vec[i], s, tq);
  }
  tq = block_sum(tq, s_red);               // (includes the barriers that publish s_row)
  for (int i = tid; i < N; i += tkThreads) p.invQt[(long long)e * N + i] = s_row[i];
  if (tid == 0)
    p.cost[e] = 0.5 * s_logdet + 0.5 * tq + 0.5 * N * 1.8378770664093453;   // log(2 pi)

  // ---- gradient --------------------------------------------------------------------------
  double s1[tkMaxD], s2[tkMaxD];
#pragma unroll
  for (int d = 0; d < tkMaxD; ++d) s1[d] = s2[d] = 0.0;
  double sz1 = 0.0, sz2 = 0.0, tr = 0.0, ss = 0.0;
  for (int i = ty; i < N; i += tkSide)
    for (int j = tx; j < N; j += tkSide) {
      double dl2[tkMaxD];
      double r2 = 0.0;
#pragma unroll
      for (int d = 0; d < tkMaxD; ++d) {
        double dl = 0.0;
        if (d < D) dl = p.inputs[i * D + d] - p.inputs[j * D + d];
        dl2[d] = dl * dl;
        r2 = fma(d < D ? s_e[d] : 0.0, dl2[d], r2);
      }
      const double z = b * exp(-0.5 * r2);
      const double q = A[(long long)i * N + j];
      const double w1 = q * z, w2 = s_row[i] * s_row[j] * z;
      sz1 += w1;
      sz2 += w2;
#pragma unroll
      for (int d = 0; d < tkMaxD; ++d) {
        s1[d] = fma(w1, dl2[d], s1[d]);
        s2[d] = fma(w2, dl2[d], s2[d]);
      }
      if (i == j) {
        tr += q;
        ss = fma(s_row[i], s_row[i], ss);
      }
    }
  double* g = p.grad + (long long)e * (D + 2);
#pragma unroll
  for (int d = 0; d < tkMaxD; ++d) {
    if (d < D) {                           // D is uniform: every thread takes the same branches
      const double a1 = block_sum(s1[d], s_red);
      const double a2 = block_sum(s2[d], s_red);
      if (tid == 0) g[d] = s_e[d] * (a2 - a1) / 4.0;
    }
  }
  sz1 = block_sum(sz1, s_red);
  sz2 = block_sum(sz2, s_red);
  tr = block_sum(tr, s_red);
  ss = block_sum(ss, s_red);
  if (tid == 0) {
    g[D] = 0.5 * sz1 - 0.5 * sz2;
    g[D + 1] = 0.5 * tr * noise - 0.5 * ss * noise;
  }
}

}  // namespace gpk
