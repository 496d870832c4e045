// Micro-benchmark (fp32 twin of mfma_f64_probe.hip): v_mfma_f32_16x16x4_f32 and v_fma_f32
// alone and together -- do they share an issue pipe as the fp64 pair does?
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_probe.hip -o tools/mfma_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ unsigned long long g_clk[4];
__device__ inline void stamp(int which) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk[which] = __builtin_amdgcn_s_memtime();
    g_clk[which + 2] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int NACC, int VALU_PER_MFMA, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(float* out, int iters, float seed) {
  stamp(0);
  f32x4 acc[NACC > 0 ? NACC : 1];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  float a = seed + threadIdx.x * 1e-6f, b = seed * 0.5f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (NACC > 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < VALU_PER_MFMA; ++k) v[k & 15] = fmaf(v[k & 15], a, b);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = fmaf(v[k], a, b);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  stamp(1);
}

template <typename K>
static int time_it(const char* name, K kern, int blocks, int threads, int iters, double flop_per_thread_iter, float* dout) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, iters, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, iters, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double flops = 3.0 * blocks * (double)threads * iters * flop_per_thread_iter;
  unsigned long long clk[4];
  CHECK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
  double ghz = (double)(clk[1] - clk[0]) / (double)(clk[3] - clk[2]) * 0.1;
  printf("%-46s waves/SIMD=%d %8.3f ms %7.2f TFLOP/s clock %.3f GHz (%.1f%% of 256CUx4x64 flop/clk)\n", name,
         threads / 256 * (blocks / 256), ms / 3, flops / (ms * 1e-3) / 1e12, ghz,
         100.0 * flops / (ms * 1e-3) / (256.0 * 4 * 64 * ghz * 1e9));
  return 0;
}

int main() {
  float* dout; CHECK(hipMalloc(&dout, sizeof(float) * 256 * 1024 * 4));
  const int iters = 20000;
  const double mf = 8 * 32.0;   // 16x16x4 MFMA = 2048 flop per wave = 32 per thread
  for (int wps = 1; wps <= 3; ++wps) {
    int blocks = 256 * wps;     // 256-thread blocks, wps per CU -> wps waves per SIMD
    time_it("mfma f32, 1 dependent accumulator", probe<1, 0, 256>, blocks, 256, iters, mf, dout);
    time_it("mfma f32, 2 accumulators", probe<2, 0, 256>, blocks, 256, iters, mf * 2, dout);
    time_it("v_fma_f32 only (16 chains)", probe<0, 0, 256>, blocks, 256, iters, 8 * 16 * 2.0, dout);
    time_it("mfma(1 acc) + 4 v_fma_f32 each [mfma flops]", probe<1, 4, 256>, blocks, 256, iters, mf, dout);
    time_it("mfma(1 acc) + 8 v_fma_f32 each [mfma flops]", probe<1, 8, 256>, blocks, 256, iters, mf, dout);
    time_it("mfma(1 acc) + 16 v_fma_f32 each [mfma flops]", probe<1, 16, 256>, blocks, 256, iters, mf, dout);
  }
  return 0;
}
