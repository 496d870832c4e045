// Reconstruction of multivariate outputs from per-PC emulator outputs, for M test rows at once:
//
//   out[r][band] = sum_p coef[p][r] * basis[p][band]
//
// With coef = the batched predict's mean [P][M] this is MultivariateEmulator.predict's
// `fwd += pred_mu * basis_functions[i]` (gp_emulator/multivariate_gp.py:214-216); with
// coef = its gradient [P][M*D] (rows r = (m, d)) it is the Jacobian
// `deriv += grad.T * basis_functions[i]` (:218).  The reference does this on the host for ONE
// test row per call.
//
// K = P <= 16 products per output against 8 bytes written: HBM-write-bound (2101 bands x 8 B =
// 16.8 KB per row), so the kernel is organised around the stores: a thread owns BPT bands
// strided by the block size (each wave store instruction writes 64 consecutive reals = 512 B
// contiguous), keeps its P x BPT basis values in registers for the whole launch, and the
// coefficient rows are staged through LDS once per workgroup and read as broadcasts.
#pragma once
#include <hip/hip_runtime.h>
#include "gp_predict_kernel.hpp"

namespace gpk {

constexpr int rkRows = 64;                     // coefficient rows staged per work item

// 16 bytes per lane per store (1 KiB contiguous per wave instruction); output rows start on
// a multiple of sizeof(T) only (B = 2101 is odd), so the vector type is declared with element
// alignment and the stores are unaligned dwordx4 (legal on global memory).
template <typename T> struct RecVec;
template <> struct RecVec<double> {
  static constexpr int kVec = 2;
  typedef double type __attribute__((ext_vector_type(2), aligned(8)));
};
template <> struct RecVec<float> {
  static constexpr int kVec = 4;
  typedef float type __attribute__((ext_vector_type(4), aligned(4)));
};

template <typename T>
struct ReconArgs {
  const T* basis;   // [P][B]
  const T* coef;    // [P][R]
  T* out;           // [R][B]
  long long R;
  int P, B;
};

// Geometry: rkThreads threads per workgroup, rkNV 16-byte vectors per thread per row, i.e. a
// workgroup covers rkThreads * rkNV * VEC consecutive bands of every row it handles.
//   (256, 2): 1024 fp64 / 2048 fp32 bands -- rows wider than that are cut into chunks;
//   (512, 3): 3072 fp64 / 6144 fp32 bands -- a 2101-band row is ONE contiguous 16.8 KB run
//             per workgroup and consecutive rows continue it (pure streaming for DRAM).
template <typename T, int rkThreads, int rkNV>
__host__ __device__ constexpr int recon_bands_per_wg() { return rkThreads * rkNV * RecVec<T>::kVec; }

template <typename T, int PT, int rkThreads, int rkNV>
__global__ __launch_bounds__(rkThreads) void reconstruct_kernel(ReconArgs<T> p) {
  constexpr int VEC = RecVec<T>::kVec;
  constexpr int BW = recon_bands_per_wg<T, rkThreads, rkNV>();
  typedef typename RecVec<T>::type vec_t;
  __shared__ __attribute__((aligned(16))) T s_c[rkRows][PT];
  const int tid = threadIdx.x;
  const int n_chunks = (p.B + BW - 1) / BW;
  const long long n_tiles = (p.R + rkRows - 1) / rkRows;
  const long long n_items = n_tiles * n_chunks;
  int cur_chunk = -1;
  T bs[PT][rkNV][VEC];

  for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int chunk = (int)(item % n_chunks);
    const long long r0 = (item / n_chunks) * rkRows;
    // vector v of this thread covers bands [band0(v), band0(v) + VEC)
    const int base = chunk * BW + tid * VEC;
    if (chunk != cur_chunk) {     // this thread's basis values for the chunk
      cur_chunk = chunk;
#pragma unroll
      for (int q = 0; q < PT; ++q)
#pragma unroll
        for (int v = 0; v < rkNV; ++v)
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const int band = base + v * rkThreads * VEC + k;
            bs[q][v][k] = (q < p.P && band < p.B) ? p.basis[(long long)q * p.B + band] : T(0);
          }
    }
    __syncthreads();              // previous item's readers of s_c are done
    for (int e = tid; e < rkRows * PT; e += rkThreads) {
      const int q = e / rkRows, i = e % rkRows;     // consecutive threads -> consecutive rows
      s_c[i][q] = (q < p.P && r0 + i < p.R) ? p.coef[(long long)q * p.R + r0 + i] : T(0);
    }
    __syncthreads();
    const int rows = (int)((p.R - r0 < rkRows) ? (p.R - r0) : rkRows);
    // bands this WAVE covers with vector v start at wbase + v * rkThreads * VEC; a wave
    // whose bands all lie beyond B (the last, partial chunk) skips the vector entirely
    const int wbase = chunk * BW + (tid & ~63) * VEC;
    for (int i = 0; i < rows; ++i) {
      T c[PT];
#pragma unroll
      for (int q = 0; q < PT; ++q) c[q] = s_c[i][q];
      T* orow = p.out + (r0 + i) * p.B;
#pragma unroll
      for (int v = 0; v < rkNV; ++v) {
        if (__builtin_amdgcn_readfirstlane(wbase + v * rkThreads * VEC) >= p.B) continue;
        const int band = base + v * rkThreads * VEC;
        vec_t acc;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          T a = T(0);
#pragma unroll
          for (int q = 0; q < PT; ++q) a = fma(c[q], bs[q][v][k], a);
          acc[k] = a;
        }
        if (band + VEC <= p.B) {
          *reinterpret_cast<vec_t*>(orow + band) = acc;
        } else {
#pragma unroll
          for (int k = 0; k < VEC; ++k)
            if (band + k < p.B) orow[band + k] = acc[k];
        }
      }
    }
  }
}

}  // namespace gpk
