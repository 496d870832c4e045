// Micro-benchmark: what does one MI355X actually sustain on v_mfma_f64_16x16x4_f64 and on
// v_fma_f64, alone and together?  Sets the `peak` used in bench.py's roofline (the local
// guides list no fp64 row).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// shader clock during the kernel = delta s_memtime / delta s_memrealtime x 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6); written by lane 0 of block 0
__device__ unsigned long long g_clk[4];
__device__ inline void stamp(int which) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk[which] = __builtin_amdgcn_s_memtime();
    g_clk[which + 2] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int NACC, int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void probe(double* out, int iters, double seed) {
  stamp(0);
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{seed, seed, seed, seed};
  double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
  double v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < VALU_PER_MFMA; ++k) v[k & 7] = fma(v[k & 7], a, b);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  stamp(1);
}

template <int VALU>
__global__ __launch_bounds__(256) void probe_valu(double* out, int iters, double seed) {
  stamp(0);
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = fma(v[k], a, b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  stamp(1);
}

template <typename K>
static int time_it(const char* name, K kern, int blocks, int iters, double flop_per_thread_iter, double* dout) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, iters, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dout, iters, 1.0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double flops = 5.0 * blocks * 256.0 * iters * flop_per_thread_iter;
  unsigned long long clk[4];
  CHECK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
  double ghz = (double)(clk[1] - clk[0]) / (double)(clk[3] - clk[2]) * 0.1;
  printf("%-44s blocks/CU=%d  %8.3f ms  %7.2f TFLOP/s  clock %.3f GHz  (%.1f%% of 256CUx4x32 flop/clk)\n", name,
         blocks / 256, ms / 5, flops / (ms * 1e-3) / 1e12, ghz,
         100.0 * flops / (ms * 1e-3) / (256.0 * 4 * 32 * ghz * 1e9));
  return 0;
}

int main() {
  double* dout; CHECK(hipMalloc(&dout, sizeof(double) * 256 * 256 * 8));
  const int iters = 40000;   // ~10-70 ms per launch so the clock settles
  // MFMA 16x16x4 f64 = 2048 flop per wave = 32 flop per thread
  const double mf = 8 * 32.0;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    int blocks = 256 * bpc;
    time_it("mfma f64, 1 dependent accumulator", probe<1, 0>, blocks, iters, mf * 1, dout);
    time_it("mfma f64, 2 accumulators", probe<2, 0>, blocks, iters, mf * 2, dout);
    time_it("mfma f64, 4 accumulators", probe<4, 0>, blocks, iters, mf * 4, dout);
    time_it("v_fma_f64 only (16 chains)", probe_valu<0>, blocks, iters, 8 * 16 * 2.0, dout);
    time_it("mfma(1 acc) + 4 v_fma_f64 each [mfma flops]", probe<1, 4>, blocks, iters, mf, dout);
    time_it("mfma(1 acc) + 8 v_fma_f64 each [mfma flops]", probe<1, 8>, blocks, iters, mf, dout);
    time_it("mfma(1 acc) + 12 v_fma_f64 each [mfma flops]", probe<1, 12>, blocks, iters, mf, dout);
    time_it("mfma(1 acc) + 16 v_fma_f64 each [mfma flops]", probe<1, 16>, blocks, iters, mf, dout);
    time_it("mfma(2 acc) + 12 v_fma_f64 each [mfma flops]", probe<2, 12>, blocks, iters, mf * 2, dout);
  }
  return 0;
}
