// Micro-benchmark for the Hessian kernel's schedule: how does ONE wave (and two per SIMD) fare on
//   (a) a dependent chain of v_fma_f64 (the exp / dot-product chain of phase A),
//   (b) ten independent v_mfma_f64_16x16x4_f64 back to back, then a 40-long dependent fma chain ("clustered":
//       what hipcc emits for a k-step of the pipelined kernel),
//   (c) the same work interleaved, one matrix instruction per four chain steps ("interleaved")?
// Every instruction is a volatile asm statement, so the order written here is the order issued.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pipe_mix_probe.hip -o tools/pipe_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(double* out, unsigned long long* cyc, int iters, double seed) {
  f64x4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = f64x4{seed, seed, seed, seed};
  double a = 1.0 + threadIdx.x * 1e-12, b = seed * 1e-3;
  double x = seed, y = seed * 2;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {            // one dependent chain, 40 steps
#pragma unroll
      for (int k = 0; k < 40; ++k) FMA(x);
    } else if constexpr (MODE == 1) {     // two chains, 20 steps each, alternating
#pragma unroll
      for (int k = 0; k < 20; ++k) { FMA(x); FMA(y); }
    } else if constexpr (MODE == 2) {     // 10 mfma only
#pragma unroll
      for (int i = 0; i < 10; ++i) MFMA(acc[i]);
    } else if constexpr (MODE == 3) {     // clustered: 10 mfma, then the 40-chain
#pragma unroll
      for (int i = 0; i < 10; ++i) MFMA(acc[i]);
#pragma unroll
      for (int k = 0; k < 40; ++k) FMA(x);
    } else if constexpr (MODE == 4) {     // interleaved: (mfma, 4 chain steps) x 10
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        MFMA(acc[i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) FMA(x);
      }
    } else if constexpr (MODE == 5) {     // interleaved, chain steps placed AFTER a pair of mfmas
#pragma unroll
      for (int i = 0; i < 10; i += 2) {
        MFMA(acc[i]);
        MFMA(acc[i + 1]);
#pragma unroll
        for (int k = 0; k < 8; ++k) FMA(x);
      }
    } else if constexpr (MODE == 6) {     // interleaved with 32-bit VALU instead of fp64 (does 32-bit VALU overlap an fp64 mfma?)
      float xf = (float)x;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        MFMA(acc[i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(xf) : "v"((float)a));
      }
      x = xf;
    } else if constexpr (MODE == 8) {     // 10 mfma on ONE accumulator (a dependent chain, as predict_kernel's phase B)
#pragma unroll
      for (int i = 0; i < 10; ++i) MFMA(acc[0]);
    } else if constexpr (MODE == 9) {     // 10 mfma alternating between TWO accumulators
#pragma unroll
      for (int i = 0; i < 10; ++i) MFMA(acc[i & 1]);
    } else if constexpr (MODE == 7) {     // 40 dependent v_fma_f32 alone
      float xf = (float)x;
#pragma unroll
      for (int k = 0; k < 40; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(xf) : "v"((float)a));
      x = xf;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = x + y;
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

template <int MODE>
int run(const char* name, int wg_per_cu, double* d_out, unsigned long long* d_cyc) {
  const int iters = 2000, grid = 256 * wg_per_cu;
  CHECK(hipMemset(d_cyc, 0, 8));
  probe<MODE><<<grid, 256>>>(d_out, d_cyc, 100, 1.0);     // warm up
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(d_cyc, 0, 8));
  probe<MODE><<<grid, 256>>>(d_out, d_cyc, iters, 1.0);
  CHECK(hipDeviceSynchronize());
  unsigned long long c;
  CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
  printf("%-58s waves/SIMD=%d  %8.1f cycles per iteration per wave\n", name, wg_per_cu, (double)c / (grid * 4) / iters);
  return 0;
}

int main() {
  double* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, 512 * 256 * 8));
  CHECK(hipMalloc(&d_cyc, 8));
  for (int w = 1; w <= 2; ++w) {
    if (run<0>("40-step dependent v_fma_f64 chain", w, d_out, d_cyc)) return 1;
    if (run<1>("two 20-step chains, alternating", w, d_out, d_cyc)) return 1;
    if (run<2>("10 independent mfma_f64", w, d_out, d_cyc)) return 1;
    if (run<3>("clustered: 10 mfma, then the 40-step chain", w, d_out, d_cyc)) return 1;
    if (run<4>("interleaved: (1 mfma, 4 chain steps) x 10", w, d_out, d_cyc)) return 1;
    if (run<5>("interleaved: (2 mfma, 8 chain steps) x 5", w, d_out, d_cyc)) return 1;
    if (run<6>("interleaved: (1 mfma, 4 dependent v_fma_f32) x 10", w, d_out, d_cyc)) return 1;
    if (run<7>("40-step dependent v_fma_f32 chain", w, d_out, d_cyc)) return 1;
    if (run<8>("10 mfma_f64 on ONE accumulator (dependent chain)", w, d_out, d_cyc)) return 1;
    if (run<9>("10 mfma_f64 alternating between two accumulators", w, d_out, d_cyc)) return 1;
  }
  return 0;
}
