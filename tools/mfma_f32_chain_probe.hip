// v_mfma_f32_16x16x4_f32 chains as predict_kernel<float> issues them: ONE dependent accumulator per wave against two
// and four, at 1..4 waves per SIMD (one 256..1024-thread workgroup per CU).  Cycles per matrix instruction and SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_chain_probe.hip -o tools/mfma_f32_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define MFMA(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  float a = 1.0f + threadIdx.x * 1e-7f, b = seed * 1e-3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) MFMA(acc[i % NACC]);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) atomicMax(cyc, t1 - t0);      // the workgroup's span ~ wave 0's
}

template <int NACC, int WAVES>
int run(float* d_out, unsigned long long* d_cyc) {
  const int iters = 2000, grid = 256;
  probe<NACC, WAVES><<<grid, WAVES * 64>>>(d_out, d_cyc, 50, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(d_cyc, 0, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  probe<NACC, WAVES><<<grid, WAVES * 64>>>(d_out, d_cyc, iters, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // all the SIMD's matrix instructions / the launch's wall time, at an assumed 2.4 GHz; and TFLOP/s
  const double n_mfma_per_simd = (double)iters * 16 * (WAVES / 4);
  const double tf = n_mfma_per_simd * 1024 * 2048.0 / (ms * 1e-3) / 1e12;    // 1024 SIMDs, 2048 flop per instruction
  printf("%d accumulator(s) per wave, %d wave(s) per SIMD: %7.1f TFLOP/s = %5.1f %% of 157.3   (%.1f cycles per instruction and SIMD at 2.4 GHz)\n",
         NACC, WAVES / 4, tf, 100 * tf / 157.3, ms * 1e-3 * 2.4e9 / n_mfma_per_simd);
  return 0;
}

int main() {
  float* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, 256 * 1024 * 4));
  CHECK(hipMalloc(&d_cyc, 8));
#define R(N, W) if (run<N, W>(d_out, d_cyc)) return 1;
  R(1, 4) R(2, 4) R(4, 4) R(1, 8) R(2, 8) R(1, 12) R(2, 12) R(4, 12) R(1, 16) R(2, 16)
  return 0;
}
