// Micro-benchmark for the fp32 predict kernel's schedule (the fp32 twin of pipe_mix_probe.hip): does an fp32 matrix
// instruction (v_mfma_f32_16x16x4_f32, 8 passes = 32 cycles) leave room for vector instructions
//   (a) of the SAME wave (clustered / interleaved orders), and
//   (b) of ANOTHER wave of the same SIMD (a workgroup of 8 or 12 waves: waves 0..3 issue matrix instructions only,
//       the others vector instructions only; each role is timed on its own)?
// Every instruction is a volatile asm statement, so the order written here is the order issued.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pipe_mix_probe_f32.hip -o tools/pipe_mix_probe_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(ad), "v"(bd))
#define MFMA(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  float a = 1.0f + threadIdx.x * 1e-6f, b = seed * 1e-3f;
  double ad = a, bd = b, xd = seed;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + i;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 xp[8], ap = {a, a + 1e-3f}, bp = {b, b};
  for (int i = 0; i < 8; ++i) xp[i] = f32x2{seed + i, seed - i};
  float x16[16];
  for (int i = 0; i < 16; ++i) x16[i] = seed + 2 * i;
  __shared__ float s_frag[128 * 64];
  for (int i = threadIdx.x; i < 128 * 64; i += THREADS) s_frag[i] = seed * 1e-3f * (i & 63);
  __syncthreads();
  const unsigned lds_addr = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)(reinterpret_cast<char*>(s_frag))) + (threadIdx.x & 63) * 4;
  float ring[4] = {0.f, 0.f, 0.f, 0.f};
  const int wave = threadIdx.x >> 6;
  const bool matrix_role = wave < 4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {            // one dependent chain, 40 steps
#pragma unroll
      for (int k = 0; k < 40; ++k) FMA(x[0]);
    } else if constexpr (MODE == 1) {     // 40 fmas on 8 independent chains
#pragma unroll
      for (int k = 0; k < 40; ++k) FMA(x[k & 7]);
    } else if constexpr (MODE == 2) {     // 10 independent mfma
#pragma unroll
      for (int i = 0; i < 10; ++i) MFMA(acc[i]);
    } else if constexpr (MODE == 3) {     // clustered: 10 mfma, then 40 independent fmas
#pragma unroll
      for (int i = 0; i < 10; ++i) MFMA(acc[i]);
#pragma unroll
      for (int k = 0; k < 40; ++k) FMA(x[k & 7]);
    } else if constexpr (MODE == 4) {     // interleaved: (mfma, 4 independent fmas) x 10
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        MFMA(acc[i]);
#pragma unroll
        for (int k = 0; k < 4; ++k) FMA(x[(4 * i + k) & 7]);
      }
    } else if constexpr (MODE == 5) {     // interleaved: (mfma, 7 independent fmas) x 10: 7 x 4 = 28 cycles of vector work per 32-cycle mfma
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        MFMA(acc[i]);
#pragma unroll
        for (int k = 0; k < 7; ++k) FMA(x[(7 * i + k) & 7]);
      }
    } else if constexpr (MODE == 6) {     // roles by wave: waves 0..3 ten mfma, the others 40 independent fmas
      if (matrix_role) {
#pragma unroll
        for (int i = 0; i < 10; ++i) MFMA(acc[i]);
      } else {
#pragma unroll
        for (int k = 0; k < 40; ++k) FMA(x[k & 7]);
      }
    } else if constexpr (MODE == 7) {     // roles by wave: waves 0..3 ten mfma, the others a 40-step dependent chain
      if (matrix_role) {
#pragma unroll
        for (int i = 0; i < 10; ++i) MFMA(acc[i]);
      } else {
#pragma unroll
        for (int k = 0; k < 40; ++k) FMA(x[0]);
      }
    } else if constexpr (MODE == 8) {     // 10 mfma on ONE accumulator (the predict kernel's phase B chain)
#pragma unroll
      for (int i = 0; i < 10; ++i) MFMA(acc[0]);
    } else if constexpr (MODE == 10) {    // 40 v_pk_fma_f32 on 8 independent chains (two fmas per lane and instruction)
#pragma unroll
      for (int k = 0; k < 40; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[k & 7]) : "v"(ap), "v"(bp));
    } else if constexpr (MODE == 11) {    // 40 v_pk_fma_f32, one dependent chain
#pragma unroll
      for (int k = 0; k < 40; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[0]) : "v"(ap), "v"(bp));
    } else if constexpr (MODE == 12) {    // 40 v_pk_fma_f32 with a broadcast operand (op_sel: the low half of src1 for both)
#pragma unroll
      for (int k = 0; k < 40; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(xp[k & 7]) : "v"(ap), "v"(bp));
    } else if constexpr (MODE == 13) {    // 40 v_pk_add_f32 on 8 chains
#pragma unroll
      for (int k = 0; k < 40; ++k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[k & 7]) : "v"(bp));
    } else if constexpr (MODE == 14) {    // 40 v_fma_f32 on 16 independent chains
#pragma unroll
      for (int k = 0; k < 40; ++k) FMA(x16[k & 15]);
    } else if constexpr (MODE == 20 || MODE == 21 || MODE == 22) {
      // the predict kernel's phase B in miniature: 40 matrix instructions on ONE accumulator, every A operand read from
      // LDS (a ring of 4 reads in flight, in-order returns), B operands from 8 registers.  21: the same without the LDS
      // reads (A from registers); 22: with s_setprio 1 around the matrix instructions
#pragma unroll
      for (int k = 0; k < 3; ++k)
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ring[k]) : "v"(lds_addr), "n"(k * 256));
#pragma unroll
      for (int k = 0; k < 40; ++k) {
        if constexpr (MODE != 21)
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(ring[(k + 3) & 3]) : "v"(lds_addr), "n"(((k + 3) % 128) * 256));
        if constexpr (MODE != 21) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        if constexpr (MODE == 22) asm volatile("s_setprio 1");
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(MODE == 21 ? x[k & 7] : ring[k & 3]), "v"(x[k & 7]));
        if constexpr (MODE == 22) asm volatile("s_setprio 0");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (MODE == 9) {     // roles by wave: waves 0..3 ten mfma, the others 20 dependent v_fma_f64 (the mean's double sum)
      if (matrix_role) {
#pragma unroll
        for (int i = 0; i < 10; ++i) MFMA(acc[i]);
      } else {
#pragma unroll
        for (int k = 0; k < 20; ++k) FMA64(xd);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)xd;
  for (int i = 0; i < 8; ++i) s += x[i] + xp[i][0] + xp[i][1];
  for (int i = 0; i < 16; ++i) s += x16[i];
  s += ring[0] + ring[1] + ring[2] + ring[3];
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc + (matrix_role ? 0 : 1), t1 - t0);
}

template <int MODE, int THREADS>
int run(const char* name, float* d_out, unsigned long long* d_cyc) {
  const int iters = 2000, grid = 256, waves = THREADS / 64;
  probe<MODE, THREADS><<<grid, THREADS>>>(d_out, d_cyc, 100, 1.0f);     // warm up
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(d_cyc, 0, 16));
  probe<MODE, THREADS><<<grid, THREADS>>>(d_out, d_cyc, iters, 1.0f);
  CHECK(hipDeviceSynchronize());
  unsigned long long c[2];
  CHECK(hipMemcpy(c, d_cyc, 16, hipMemcpyDeviceToHost));
  const double first = (double)c[0] / (grid * 4) / iters;
  const double rest = waves > 4 ? (double)c[1] / (grid * (waves - 4)) / iters : 0.0;
  printf("%-66s waves/SIMD=%d  waves 0-3: %7.1f   other waves: %7.1f  cycles per iteration per wave\n", name, waves / 4, first, rest);
  return 0;
}

int main() {
  float* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, 256 * 768 * 4));
  CHECK(hipMalloc(&d_cyc, 16));
#define ALLW(M, name) if (run<M, 256>(name, d_out, d_cyc) || run<M, 512>(name, d_out, d_cyc) || run<M, 768>(name, d_out, d_cyc)) return 1;
  ALLW(0, "40-step dependent v_fma_f32 chain")
  ALLW(1, "40 v_fma_f32 on 8 independent chains")
  ALLW(2, "10 independent mfma_f32_16x16x4")
  ALLW(8, "10 mfma_f32 on ONE accumulator (dependent chain)")
  ALLW(3, "clustered: 10 mfma, then 40 independent fmas")
  ALLW(4, "interleaved: (1 mfma, 4 independent fmas) x 10")
  ALLW(5, "interleaved: (1 mfma, 7 independent fmas) x 10")
  ALLW(14, "40 v_fma_f32 on 16 independent chains")
  ALLW(10, "40 v_pk_fma_f32 on 8 independent chains")
  ALLW(11, "40-step dependent v_pk_fma_f32 chain")
  ALLW(12, "40 v_pk_fma_f32, one operand broadcast by op_sel_hi")
  ALLW(13, "40 v_pk_add_f32 on 8 independent chains")
  ALLW(20, "phase B in miniature: 40 mfma, A operands from LDS (ring of 4)")
  ALLW(21, "the same, A operands from registers")
  ALLW(22, "the same as the first, s_setprio 1 around each mfma")
  if (run<6, 512>("roles: waves 0-3 ten mfma | waves 4-7 forty independent fmas", d_out, d_cyc)) return 1;
  if (run<6, 768>("roles: waves 0-3 ten mfma | waves 4-11 forty independent fmas", d_out, d_cyc)) return 1;
  if (run<7, 512>("roles: waves 0-3 ten mfma | waves 4-7 a 40-step dependent chain", d_out, d_cyc)) return 1;
  if (run<9, 512>("roles: waves 0-3 ten mfma | waves 4-7 twenty dependent v_fma_f64", d_out, d_cyc)) return 1;
  return 0;
}
