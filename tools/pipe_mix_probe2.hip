// Second micro-benchmark for the Hessian kernel's schedule (see pipe_mix_probe.hip): two waves per SIMD in ONE
// 512-thread workgroup (waves w and w + 4 share a SIMD), each half with its own role, and matrix instructions
// alone at 1..4 waves per SIMD.  Every instruction is a volatile asm statement.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pipe_mix_probe2.hip -o tools/pipe_mix_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// role: 0 = 10 mfma; 1 = 40-step chain; 2 = two 20-step chains; 3 = 10 mfma then two 20-step chains;
//       4 = 190 mfma then 38 x (two 20-step chains)  [window-sized phases];  5 = idle (exits at once)
template <int ROLE>
__device__ __forceinline__ void body(f64x4 (&acc)[10], double& x, double& y, double a, double b) {
  if constexpr (ROLE == 0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) MFMA(acc[i]);
  } else if constexpr (ROLE == 1) {
#pragma unroll
    for (int k = 0; k < 40; ++k) FMA(x);
  } else if constexpr (ROLE == 2) {
#pragma unroll
    for (int k = 0; k < 20; ++k) { FMA(x); FMA(y); }
  } else if constexpr (ROLE == 3) {
#pragma unroll
    for (int i = 0; i < 10; ++i) MFMA(acc[i]);
#pragma unroll
    for (int k = 0; k < 20; ++k) { FMA(x); FMA(y); }
  } else if constexpr (ROLE == 4) {
    for (int r = 0; r < 19; ++r) {
#pragma unroll
      for (int i = 0; i < 10; ++i) MFMA(acc[i]);
    }
    for (int r = 0; r < 19; ++r) {
#pragma unroll
      for (int k = 0; k < 20; ++k) { FMA(x); FMA(y); }
    }
  }
}

template <int RA, int RB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe(double* out, unsigned long long* cyc, int iters, double seed) {
  f64x4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = f64x4{seed, seed, seed, seed};
  double a = 1.0 + threadIdx.x * 1e-12, b = seed * 1e-3;
  double x = seed, y = seed * 2;
  const int half = (threadIdx.x >> 6) >= 4 ? 1 : 0;      // waves 4.. share SIMDs with waves 0..3
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (half == 0) {
    if constexpr (RA != 5) for (int it = 0; it < iters; ++it) body<RA>(acc, x, y, a, b);
  } else {
    if constexpr (RB != 5) for (int it = 0; it < iters; ++it) body<RB>(acc, x, y, a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = x + y;
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc + half, t1 - t0);
}

template <int RA, int RB, int WAVES>
int run(const char* name, double* d_out, unsigned long long* d_cyc, int iters = 2000) {
  const int grid = 256;
  probe<RA, RB, WAVES><<<grid, WAVES * 64>>>(d_out, d_cyc, 50, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(d_cyc, 0, 16));
  probe<RA, RB, WAVES><<<grid, WAVES * 64>>>(d_out, d_cyc, iters, 1.0);
  CHECK(hipDeviceSynchronize());
  unsigned long long c[2];
  CHECK(hipMemcpy(c, d_cyc, 16, hipMemcpyDeviceToHost));
  const int wa = WAVES >= 4 ? 4 : WAVES, wb = WAVES - wa;
  printf("%-66s first half %8.1f", name, (double)c[0] / (grid * wa) / iters);
  if (wb > 0) printf("   second half %8.1f", (double)c[1] / (grid * wb) / iters);
  printf("   cycles per iteration per wave\n");
  return 0;
}

int main() {
  double* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, 256 * 1024 * 8));
  CHECK(hipMalloc(&d_cyc, 16));
  if (run<0, 5, 4>("10 mfma, 1 wave per SIMD", d_out, d_cyc)) return 1;
  if (run<0, 0, 8>("10 mfma, 2 waves per SIMD", d_out, d_cyc)) return 1;
  if (run<0, 0, 12>("10 mfma, 3 waves per SIMD (second half = waves 4..11)", d_out, d_cyc)) return 1;
  if (run<0, 0, 16>("10 mfma, 4 waves per SIMD (second half = waves 4..15)", d_out, d_cyc)) return 1;
  if (run<0, 1, 8>("first half 10 mfma | second half 40-step chain", d_out, d_cyc)) return 1;
  if (run<0, 2, 8>("first half 10 mfma | second half two 20-step chains", d_out, d_cyc)) return 1;
  if (run<2, 2, 8>("both two 20-step chains", d_out, d_cyc)) return 1;
  if (run<3, 3, 8>("both: 10 mfma then two 20-step chains", d_out, d_cyc)) return 1;
  if (run<3, 5, 4>("alone: 10 mfma then two 20-step chains", d_out, d_cyc)) return 1;
  if (run<4, 4, 8>("both: 190 mfma then 38 x two 20-step chains (per 19 units)", d_out, d_cyc, 200)) return 1;
  if (run<4, 5, 4>("alone: 190 mfma then 38 x two 20-step chains (per 19 units)", d_out, d_cyc, 200)) return 1;
  return 0;
}
