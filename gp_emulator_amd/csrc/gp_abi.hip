// C ABI of libgp_predict_hip.so: see include/gp_predict_hip.h for the contract and the
// reference interfaces (file:line) each entry replaces.
#include "gp_predict_hip.h"

#include <hip/hip_runtime.h>

#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <atomic>
#include <mutex>
#include <new>
#include <thread>
#include <type_traits>
#include <vector>

#include <sched.h>

#include "gp_dispatch.hpp"
#include "gp_generic_kernel.hpp"
#include "gp_hessian_kernel.hpp"
#include "gp_hessian_mfma_kernel.hpp"
#include "gp_hessian_win_kernel.hpp"
#include "gp_host_pool.hpp"
#include "gp_predict_kernel.hpp"
#include "gp_reconstruct_kernel.hpp"
#include "gp_train_args.hpp"

namespace gpk {
hipError_t launch_likelihood(const TrainArgs&, int n_sets, hipStream_t);
hipError_t launch_reconstruct_f32(const ReconArgs<float>&, int wide, int cus, hipStream_t);
hipError_t launch_reconstruct_f64(const ReconArgs<double>&, int wide, int cus, hipStream_t);
hipError_t launch_few_f32(int, const PredictArgs<float>&, int, int, hipStream_t);
hipError_t launch_few_f64(int, const PredictArgs<double>&, int, int, hipStream_t);
hipError_t launch_generic_f32(const GenericArgs<float>&, int, hipStream_t);
hipError_t launch_generic_f64(const GenericArgs<double>&, int, hipStream_t);
#define GP_DECL(nb)                                                                          \
  hipError_t launch_hessm_f32_##nb(int, const HessMfmaArgs<float>&, int, hipStream_t);      \
  hipError_t launch_hessm_f64_##nb(int, const HessMfmaArgs<double>&, int, hipStream_t);
GP_FOR_EACH_KERNEL_NB(GP_DECL)
#undef GP_DECL
hipError_t launch_hessian_f32(int, const HessianArgs<float>&, int, hipStream_t);
hipError_t launch_hessian_f64(int, const HessianArgs<double>&, int, hipStream_t);
#define GP_DECL(nk)                                                                         \
  hipError_t launch_predict_f32_##nk(int, const PredictArgs<float>&, int, hipStream_t);     \
  hipError_t launch_predict_f64_##nk(int, const PredictArgs<double>&, int, hipStream_t);
GP_FOR_EACH_KERNEL_NK(GP_DECL)
#undef GP_DECL
}  // namespace gpk

// ------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// No C++ exception may cross the C boundary: the exported functions that allocate host memory
// (packing buffers, cache keys) run their bodies through this.
template <typename F>
static int guarded(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(GP_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& ex) {
    return fail(GP_ERR_INVALID, "%s", ex.what());
  } catch (...) {
    return fail(GP_ERR_INVALID, "unknown C++ exception");
  }
}

#define HIP_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return fail(GP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                  __FILE__, __LINE__);                                                \
  } while (0)

struct gp_model {
  int device;
  int dtype;
  int n_train, n_inputs;
  int kernel_d, kernel_nb;
  int kernel_nk;                           // k-steps the predict kernel is compiled for (0: general-shape kernel)
  int n_emulators;                         // > 1: batched emulators sharing inputs/test rows
  long long xa_stride, frags_stride, sd_stride;   // elements between emulators
  void* d_xa;
  void* d_frags;
  void* d_sd;
  // the scale sqrt(e_d) and centre c_d the kernel applies to test rows, as doubles (emulator 0):
  // the host applies them itself, in double, when it stages float64 rows for a float32 model
  std::vector<double> scale_host, centre_host;
  // Hessian on the matrix core (gp_hessian_mfma_kernel.hpp): the constant products
  // x''_id x''_id2 in fragment order, built on the first Hessian call from a host copy of the
  // scaled rows (double, [16 * kernel_nb slots][kernel_d])
  std::vector<double> xs_host;
  std::mutex h_mutex;
  void* d_pfrags;
};

// Host-pointer path (predict_host): slabs of the caller's arrays flow through kPipeSlots slots,
// each with its own stream, pinned staging buffers and device buffers.
constexpr int kPipeSlotsMax = 6;
// slots in use (GP_HOST_SLOTS, default 3)
static int pipe_slots() {
  static const int n = [] {
    const char* ev = getenv("GP_HOST_SLOTS");
    const int v = ev ? atoi(ev) : 3;
    return v < 2 ? 2 : (v > kPipeSlotsMax ? kPipeSlotsMax : v);
  }();
  return n;
}
#define kPipeSlots pipe_slots()
struct gp_pipe {
  bool ready = false;
  hipStream_t stream[kPipeSlotsMax] = {};
  hipEvent_t done[kPipeSlotsMax] = {};
  // one queue per copy direction (see run_slab_pipeline): every slab's rows go up on `up`, every slab's results
  // come down on `down`, the slot's own stream carries the kernel between two events
  hipStream_t up = nullptr, down = nullptr;
  hipEvent_t in_there[kPipeSlotsMax] = {}, computed[kPipeSlotsMax] = {};
  void* stage_in[kPipeSlotsMax] = {};
  void* stage_out[kPipeSlotsMax] = {};
  void* dev[kPipeSlotsMax] = {};

  size_t stage_in_bytes = 0, stage_out_bytes = 0, dev_bytes = 0;
  std::unique_ptr<gph::ThreadPool> pool;
};

// predict_wrap re-sends the emulator's constants with every block (as the reference's boundary
// does, GaussianProcess.py:313-316).  The context remembers the last few packed models together
// with a copy of the host arrays they were made from; a call whose constants compare equal
// byte for byte reuses the model instead of packing, allocating and uploading again.
struct gp_cached_model {
  gp_model* model = nullptr;
  int host_dtype = 0, compute_dtype = 0, n_train = 0, n_inputs = 0, theta_size = 0;
  bool with_invq = false;
  std::vector<char> key;                   // expX | inputs | invQt | invQ, as given
  unsigned long long stamp = 0;
};
constexpr int kModelCacheSlots = 4;

static const int kTicketSlots = 256;   // launches that may be in flight at once on a context (the pipeline has 3 slots)

struct gp_ctx {
  int device;
  hipStream_t stream;
  int compute_units;
  // grow-only device scratch (likelihood batch)
  void* scratch;
  size_t scratch_bytes;
  void* dbg;   // diagnostic (GP_STAMPS) builds: device buffer for segment cycle sums
  // item counters of kernels that draw their work items (hessian_win_kernel): a ring of device words, all 0
  // between launches (the kernel that uses one puts it back to 0), one per launch in flight
  unsigned* tickets = nullptr;
  std::atomic<unsigned> ticket_next{0};
  gp_pipe pipe;
  gp_cached_model cache[kModelCacheSlots];
  unsigned long long cache_clock = 0;
};

struct gp_event {
  int device;
  hipEvent_t ev;
};

static const int kKernelD[] = {
#define GP_V(d) d,
    GP_FOR_EACH_KERNEL_D(GP_V)
#undef GP_V
};
static const int kKernelNK[] = {
#define GP_V(d) d,
    GP_FOR_EACH_KERNEL_NK(GP_V)
#undef GP_V
};

// Kernel choice.  *knk > 0: the fused MFMA kernel predict_kernel<T, *kd, *knk> (*knk k-steps of
// 4 training points; the packed images hold *knb = ceil(*knk / 4) blocks of 16).
// *knb == 0: the general-shape kernel (gp_generic_kernel.hpp); *kd is then the padded row
// dimension (a compiled kernel D when n_inputs <= 16, so the Hessian kernel can share the
// packed rows; n_inputs itself otherwise).
static int pick_kernel(int n_train, int n_inputs, int* kd, int* knb, int* knk = nullptr) {
  *kd = *knb = -1;
  int nk = -1;
  for (int d : kKernelD)
    if (d >= n_inputs) { *kd = d; break; }
  static const bool whole_blocks = [] {      // A/B switch: kernels of whole 16-blocks only
    const char* ev = getenv("GP_NO_KSKIP");
    return ev && atoi(ev) != 0;
  }();
  const int need = (n_train + 3) / 4;
  for (int k : kKernelNK)
    if (k >= need && !(whole_blocks && k % 4 != 0)) { nk = k; break; }
  if (nk > 0) *knb = (nk + 3) / 4;
  if (knk) *knk = nk;
  if (*kd > 0 && *knb > 0) return GP_OK;
  if (n_train <= gpk::gkMaxN && n_inputs <= gpk::gkMaxD) {
    if (*kd < 0) *kd = n_inputs;
    *knb = 0;
    if (knk) *knk = 0;
    return GP_OK;
  }
  return fail(GP_ERR_UNSUPPORTED,
              "shape outside the compiled kernel set: n_train=%d (max %d), n_inputs=%d (max %d)",
              n_train, gpk::gkMaxN, n_inputs, gpk::gkMaxD);
}
static inline int rows_padded(int n_train, int knb) { return knb > 0 ? 16 * knb : 16 * ((n_train + 15) / 16); }
static inline int row_stride_of(int kd) { return gpk::row_stride(kd); }

// ------------------------------------------------------------------------------------
// host-side packing (double arithmetic, one rounding to T at the end)
// ------------------------------------------------------------------------------------
template <typename T, typename TH = T>
static int pack_model(const TH* expX, const TH* inputs, const TH* invQt, const TH* invQ, int N,
                      int D, int theta_size, T* xa, T* frags, T* sd, T* b) {
  if (!expX || !inputs || !invQt || !xa || !sd || !b || (invQ && !frags))
    return fail(GP_ERR_INVALID, "null pointer");
  if (N <= 0 || D <= 0) return fail(GP_ERR_INVALID, "n_train and n_inputs must be positive");
  if (theta_size < D + 1)
    return fail(GP_ERR_INVALID, "theta_size=%d < n_inputs+1=%d", theta_size, D + 1);
  int kd, knb;
  int rc = pick_kernel(N, D, &kd, &knb);
  if (rc) return rc;
  const int DS = row_stride_of(kd);
  const int NP = rows_padded(N, knb);
  // sqrt(e_d): the reference scales both point sets by sqrt(expX[:D]) before cdist
  // (GaussianProcess.py:232-233; the CUDA path takes the sqrt on the host too,
  // _gpu_predict.cpp:135-140).  Both sets are also shifted by the training mean c_d first:
  // distances are unchanged and the kernel's expansion
  //   -|x''-t''|^2/2 = h_i + g + x''.t''   cancels less the smaller |x''|, |t''| are.
  std::vector<double> sdd(kd, 0.0), ctr(kd, 0.0);
  for (int d = 0; d < D; ++d) {
    sdd[d] = std::sqrt((double)expX[d]);
    double sum = 0.0;
    for (int i = 0; i < N; ++i) sum += (double)inputs[(size_t)i * D + d];
    ctr[d] = (double)(T)(sum / N);     // representable in T: the kernel subtracts it in T
  }
  for (int d = 0; d < kd; ++d) {
    sd[d] = (T)sdd[d];
    sd[kd + d] = (T)ctr[d];
  }
  *b = (T)expX[D];
  sd[2 * kd] = (T)expX[D];
  const double lnb = std::log((double)expX[D]);
  std::memset(xa, 0, sizeof(T) * (size_t)NP * DS);
  // Training point i lives in slot point_slot(i) of the packed images when the fused kernel
  // runs (gp_predict_kernel.hpp: a partly filled last block then fills whole k-steps first);
  // the general-shape kernel keeps the natural order.
  for (int i = 0; i < N; ++i) {
    const size_t row = (size_t)(knb > 0 ? gpk::point_slot<T>(i) : i) * DS;
    double n2 = 0.0;
    for (int d = 0; d < D; ++d) {
      // the kernel works with the ROUNDED x'' (type T), so h must be built from it too
      const T xr = (T)((double)sd[d] * ((double)inputs[(size_t)i * D + d] - ctr[d]));
      xa[row + d] = xr;
      n2 += (double)xr * (double)xr;
    }
    xa[row + kd] = (T)invQt[i];
    xa[row + kd + 1] = (T)(lnb - 0.5 * n2);
  }
  // S' in fragment order: fragment (I >= J, s), lane l holds
  //   S'[slot i = 16 I + own_sub(s, l >> 4)][slot j = 16 J + (l & 15)]
  // with S'_IJ = M_IJ + M_JI^T for I > J and M_JJ on the diagonal, so that
  //   k^T M k = sum_J sum_{I>=J} k_I^T S'_IJ k_J        for ANY matrix M.
  if (!invQ) return GP_OK;   // Hessian-only model: no variance operand
  if (knb == 0) {            // general-shape kernel: invQ as given
    for (size_t q = 0; q < (size_t)N * N; ++q) frags[q] = (T)invQ[q];
    return GP_OK;
  }
  const int nfp = gpk::frag_count_padded(knb, gpk::Geo<T>::kChunk);
  std::memset(frags, 0, sizeof(T) * (size_t)nfp * 64);
  for (int J = 0; J < knb; ++J)
    for (int I = J; I < knb; ++I)
      for (int s = 0; s < 4; ++s) {
        T* f = frags + (size_t)gpk::frag_index(I, J, s, knb) * 64;
        for (int l = 0; l < 64; ++l) {
          const int i = gpk::slot_point<T>(gpk::own_index<T>(I, s, l >> 4));
          const int j = gpk::slot_point<T>(16 * J + (l & 15));
          if (i >= N || j >= N) continue;
          double v = (double)invQ[(size_t)i * N + j];
          if (I > J) v += (double)invQ[(size_t)j * N + i];
          f[l] = (T)v;
        }
      }
  return GP_OK;
}

template <typename T>
static hipError_t launch(int knk, int kd, const gpk::PredictArgs<T>& a, int grid, hipStream_t s);
template <>
hipError_t launch<float>(int knk, int kd, const gpk::PredictArgs<float>& a, int grid, hipStream_t s) {
  switch (knk) {
#define GP_CASE(nk) case nk: return gpk::launch_predict_f32_##nk(kd, a, grid, s);
    GP_FOR_EACH_KERNEL_NK(GP_CASE)
#undef GP_CASE
  }
  return hipErrorInvalidValue;
}
template <>
hipError_t launch<double>(int knk, int kd, const gpk::PredictArgs<double>& a, int grid, hipStream_t s) {
  switch (knk) {
#define GP_CASE(nk) case nk: return gpk::launch_predict_f64_##nk(kd, a, grid, s);
    GP_FOR_EACH_KERNEL_NK(GP_CASE)
#undef GP_CASE
  }
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------
// exported functions
// ------------------------------------------------------------------------------------
extern "C" {

const char* gp_last_error_string(void) { return g_err; }
const char* gp_version_string(void) { return "gp_predict_hip 0.1 (gfx950)"; }

int gp_device_count(int* count) {
  if (!count) return fail(GP_ERR_INVALID, "null pointer");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(GP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return GP_OK;
}

int gp_ctx_create(int device, gp_ctx** out) {
  if (!out) return fail(GP_ERR_INVALID, "null pointer");
  *out = nullptr;
  int n = 0;
  int rc = gp_device_count(&n);
  if (rc) return rc;
  if (n <= 0) return fail(GP_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n) return fail(GP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  gp_ctx* c = new (std::nothrow) gp_ctx();
  if (!c) return fail(GP_ERR_NOMEM, "out of host memory");
  c->device = device;
  c->compute_units = prop.multiProcessorCount;
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  c->dbg = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(GP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  e = hipMalloc((void**)&c->tickets, 4 * kTicketSlots * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(c->tickets, 0, 4 * kTicketSlots * sizeof(unsigned));
  if (e != hipSuccess) {
    if (c->tickets) (void)hipFree(c->tickets);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return fail(GP_ERR_HIP, "context item counters: %s", hipGetErrorString(e));
  }
  *out = c;
  return GP_OK;
}

int gp_ctx_destroy(gp_ctx* ctx) {
  if (!ctx) return GP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& c : ctx->cache)
    if (c.model) gp_model_destroy(c.model);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->tickets) (void)hipFree(ctx->tickets);
  gp_pipe& pp = ctx->pipe;
  pp.pool.reset();
  for (int k = 0; k < kPipeSlots; ++k) {
    if (pp.stream[k]) (void)hipStreamSynchronize(pp.stream[k]);
    if (pp.stage_in[k]) (void)hipHostFree(pp.stage_in[k]);
    if (pp.stage_out[k]) (void)hipHostFree(pp.stage_out[k]);
    if (pp.dev[k]) (void)hipFree(pp.dev[k]);
    if (pp.done[k]) (void)hipEventDestroy(pp.done[k]);
    if (pp.in_there[k]) (void)hipEventDestroy(pp.in_there[k]);
    if (pp.computed[k]) (void)hipEventDestroy(pp.computed[k]);
    if (pp.stream[k]) (void)hipStreamDestroy(pp.stream[k]);
  }
  if (pp.up) { (void)hipStreamSynchronize(pp.up); (void)hipStreamDestroy(pp.up); }
  if (pp.down) { (void)hipStreamSynchronize(pp.down); (void)hipStreamDestroy(pp.down); }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return GP_OK;
}

int gp_ctx_set_debug_buffer(gp_ctx* ctx, void* d_buffer) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  ctx->dbg = d_buffer;
  return GP_OK;
}

int gp_ctx_synchronize(gp_ctx* ctx) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return GP_OK;
}

int gp_ctx_device_info(gp_ctx* ctx, int* compute_units, int64_t* hbm_bytes, char* name, int name_len) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return GP_OK;
}

int gp_frag_index(int kernel_nb, int I, int J, int s) {
  if (kernel_nb <= 0 || J < 0 || I < J || I >= kernel_nb || s < 0 || s > 3) return -1;
  return gpk::frag_index(I, J, s, kernel_nb);
}

int gp_pack_sizes(int dtype, int n_train, int n_inputs, int* kernel_d, int* kernel_nb,
                  int64_t* xa_len, int64_t* frags_len) {
  int kd, knb;
  int rc = pick_kernel(n_train, n_inputs, &kd, &knb);
  if (rc) return rc;
  if (kernel_d) *kernel_d = kd;
  if (kernel_nb) *kernel_nb = knb;
  if (xa_len) *xa_len = (int64_t)rows_padded(n_train, knb) * row_stride_of(kd);
  if (frags_len)
    *frags_len = knb > 0 ? (int64_t)gpk::frag_count_padded(
                               knb, dtype == GP_F64 ? gpk::Geo<double>::kChunk : gpk::Geo<float>::kChunk) * 64
                         : (int64_t)n_train * n_train;
  return GP_OK;
}

int gp_pack_model_f64(const double* expX, const double* inputs, const double* invQt,
                      const double* invQ, int n_train, int n_inputs, int theta_size,
                      double* xa, double* frags, double* sd, double* b) {
  return guarded([&] { return pack_model<double>(expX, inputs, invQt, invQ, n_train, n_inputs, theta_size, xa, frags, sd, b); });
}
int gp_pack_model_f32(const float* expX, const float* inputs, const float* invQt,
                      const float* invQ, int n_train, int n_inputs, int theta_size,
                      float* xa, float* frags, float* sd, float* b) {
  return guarded([&] { return pack_model<float>(expX, inputs, invQt, invQ, n_train, n_inputs, theta_size, xa, frags, sd, b); });
}

}  // extern "C"

// Pack and upload E emulators that share the training inputs (E = 1: the plain case).
// expX is [E][theta_size], invQt [E][N], invQ [E][N][N] (or null: Hessian-only model).
template <typename T, typename TH = T>
static int model_create(gp_ctx* ctx, int E, const TH* expX, const TH* inputs, const TH* invQt,
                        const TH* invQ, int N, int D, int theta_size, gp_model** out) {
  if (!ctx || !out) return fail(GP_ERR_INVALID, "null context or output");
  *out = nullptr;
  if (E <= 0) return fail(GP_ERR_INVALID, "n_emulators must be positive");
  int kd, knb, knk;
  int64_t xa_len, fr_len;
  int rc = gp_pack_sizes(sizeof(T) == 8 ? GP_F64 : GP_F32, N, D, &kd, &knb, &xa_len, &fr_len);
  if (rc) return rc;
  (void)pick_kernel(N, D, &kd, &knb, &knk);
  const int64_t sd_len = 2 * kd + 1;
  if (!invQ) fr_len = 0;
  HIP_TRY(hipSetDevice(ctx->device));
  gp_model* m = new (std::nothrow) gp_model();
  if (!m) return fail(GP_ERR_NOMEM, "out of host memory");
  m->device = ctx->device;
  m->dtype = sizeof(T) == 8 ? GP_F64 : GP_F32;
  m->n_train = N;
  m->n_inputs = D;
  m->kernel_d = kd;
  m->kernel_nb = knb;
  m->kernel_nk = knk;
  m->n_emulators = E;
  m->xa_stride = xa_len;
  m->frags_stride = fr_len;
  m->sd_stride = sd_len;
  m->d_xa = m->d_frags = m->d_sd = nullptr;
  m->d_pfrags = nullptr;
  hipError_t e = hipMalloc(&m->d_xa, sizeof(T) * xa_len * E);
  if (e == hipSuccess && invQ) e = hipMalloc(&m->d_frags, sizeof(T) * fr_len * E);
  if (e == hipSuccess) e = hipMalloc(&m->d_sd, sizeof(T) * sd_len * E);
  // pack one emulator at a time into a bounded host staging buffer, copy, repeat
  std::vector<T> xa(xa_len), fr(fr_len), sd(sd_len);
  for (int k = 0; k < E && e == hipSuccess && rc == GP_OK; ++k) {
    T b;
    rc = pack_model<T, TH>(expX + (size_t)k * theta_size, inputs, invQt + (size_t)k * N,
                       invQ ? invQ + (size_t)k * N * N : nullptr, N, D, theta_size,
                       xa.data(), invQ ? fr.data() : nullptr, sd.data(), &b);
    if (rc) break;
    if (k == 0) {              // what the kernel applies to test rows, as doubles
      m->scale_host.resize(D);
      m->centre_host.resize(D);
      for (int d = 0; d < D; ++d) {
        m->scale_host[d] = (double)sd[d];
        m->centre_host[d] = (double)sd[kd + d];
      }
    }
    if (E == 1 && knb > 0) {   // the rounded x'' the kernels see (by slot), for the Hessian's product matrix
      const int DSk = row_stride_of(kd);
      const int NPk = rows_padded(N, knb);
      m->xs_host.resize((size_t)NPk * kd);
      for (int i = 0; i < NPk; ++i)
        for (int d = 0; d < kd; ++d) m->xs_host[(size_t)i * kd + d] = (double)xa[(size_t)i * DSk + d];
    }
    e = hipMemcpyAsync((T*)m->d_xa + (size_t)k * xa_len, xa.data(), sizeof(T) * xa_len, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && invQ)
      e = hipMemcpyAsync((T*)m->d_frags + (size_t)k * fr_len, fr.data(), sizeof(T) * fr_len, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync((T*)m->d_sd + (size_t)k * sd_len, sd.data(), sizeof(T) * sd_len, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // staging buffers are reused
  }
  if (rc) { gp_model_destroy(m); return rc; }
  if (e != hipSuccess) {
    gp_model_destroy(m);
    return fail(GP_ERR_HIP, "model upload: %s", hipGetErrorString(e));
  }
  *out = m;
  return GP_OK;
}

template <typename T>
static hipError_t launch_generic(const gpk::GenericArgs<T>& a, int grid, hipStream_t s);
template <>
hipError_t launch_generic<float>(const gpk::GenericArgs<float>& a, int grid, hipStream_t s) {
  return gpk::launch_generic_f32(a, grid, s);
}
template <>
hipError_t launch_generic<double>(const gpk::GenericArgs<double>& a, int grid, hipStream_t s) {
  return gpk::launch_generic_f64(a, grid, s);
}

template <typename T>
static int predict_device(gp_ctx* ctx, const gp_model* m, const void* d_testing, void* d_mu,
                          void* d_var, void* d_deriv, int64_t M, int layout,
                          hipStream_t stream = nullptr, bool rows_prescaled = false) {
  if (!stream) stream = ctx->stream;
  if (m->kernel_nb == 0) {   // general-shape kernel
    if (m->n_emulators != 1)
      return fail(GP_ERR_UNSUPPORTED, "batched emulators need n_train <= %d and n_inputs <= %d",
                  16 * GP_MAX_KERNEL_NB, GP_MAX_KERNEL_D);
    gpk::GenericArgs<T> g;
    g.xa = (const T*)m->d_xa;
    g.invQ = (const T*)m->d_frags;
    g.sd = (const T*)m->d_sd;
    g.testing = (const T*)d_testing;
    g.mu = (T*)d_mu;
    g.var = (T*)d_var;
    g.deriv = (T*)d_deriv;
    g.M = M;
    g.N = m->n_train;
    g.D = m->n_inputs;
    g.ds = row_stride_of(m->kernel_d);
    g.acol = m->kernel_d;
    g.dk = m->kernel_d;
    g.deriv_row_major = layout == GP_DERIV_ROWMAJOR;
    int64_t tiles = (M + 15) / 16;
    int64_t grid = (int64_t)ctx->compute_units * 4;
    if (grid > tiles) grid = tiles;
    hipError_t e = launch_generic<T>(g, (int)grid, stream);
    if (e != hipSuccess) return fail(GP_ERR_HIP, "generic kernel launch: %s", hipGetErrorString(e));
    return GP_OK;
  }
  gpk::PredictArgs<T> a;
  a.xa = (const T*)m->d_xa;
  a.frags = (const T*)m->d_frags;
  a.sd = (const T*)m->d_sd;
  a.testing = (const T*)d_testing;
  a.mu = (T*)d_mu;
  a.var = (T*)d_var;
  a.deriv = (T*)d_deriv;
  a.M = M;
  a.d_actual = m->n_inputs;
  a.deriv_row_major = layout == GP_DERIV_ROWMAJOR;
  a.n_emulators = m->n_emulators;
  a.xa_stride = m->xa_stride;
  a.frags_stride = m->frags_stride;
  a.sd_stride = m->sd_stride;
  a.dbg = (unsigned long long*)ctx->dbg;
  a.rows_prescaled = rows_prescaled ? 1 : 0;
  // Few rows (one state vector at a time): the latency form, a workgroup per 16-row tile with the
  // tile's work shared by its waves (gp_predict_few_kernel.hpp), while every tile still gets a
  // workgroup of its own in one round of the chip.  GP_NO_FEW=1: always the throughput kernel.
  const char* few_ev = getenv("GP_NO_FEW");          // read per call: the tests run both kernels in one process
  const bool no_few = few_ev && atoi(few_ev) != 0;
  const int64_t tiles = (M + gpk::kTile - 1) / gpk::kTile * m->n_emulators;
  if (!no_few && tiles <= 2 * (int64_t)ctx->compute_units) {
    hipError_t e;
    if constexpr (sizeof(T) == 8) e = gpk::launch_few_f64(m->kernel_d, a, m->kernel_nb, (int)tiles, stream);
    else e = gpk::launch_few_f32(m->kernel_d, a, m->kernel_nb, (int)tiles, stream);
    if (e != hipSuccess) return fail(GP_ERR_HIP, "kernel launch (few rows): %s", hipGetErrorString(e));
    return GP_OK;
  }
  constexpr int kRowsPerWG = gpk::Geo<T>::kRowsPerWG;
  const int64_t groups = (M + kRowsPerWG - 1) / kRowsPerWG * m->n_emulators;
  if ((M + kRowsPerWG - 1) / kRowsPerWG > 0x7fffffffLL)
    return fail(GP_ERR_INVALID, "n_predict too large for one launch");
  // persistent grid: the kernel's occupancy (2 waves per SIMD), grid-stride over work items
  int64_t grid = (int64_t)ctx->compute_units * gpk::Geo<T>::kWGPerCU;
  if (grid > groups) grid = groups;
  hipError_t e = launch<T>(m->kernel_nk, m->kernel_d, a, (int)grid, stream);
  if (e != hipSuccess) return fail(GP_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return GP_OK;
}

template <typename T>
static hipError_t launch_hessian(int kd, const gpk::HessianArgs<T>& a, int grid, hipStream_t s);
template <>
hipError_t launch_hessian<float>(int kd, const gpk::HessianArgs<float>& a, int grid, hipStream_t s) {
  return gpk::launch_hessian_f32(kd, a, grid, s);
}
template <>
hipError_t launch_hessian<double>(int kd, const gpk::HessianArgs<double>& a, int grid, hipStream_t s) {
  return gpk::launch_hessian_f64(kd, a, grid, s);
}

template <typename T>
static hipError_t launch_hessm(int knb, int kd, const gpk::HessMfmaArgs<T>& a, int grid, hipStream_t s);
template <>
hipError_t launch_hessm<float>(int knb, int kd, const gpk::HessMfmaArgs<float>& a, int grid, hipStream_t s) {
  switch (knb) {
#define GP_CASE(nb) case nb: return gpk::launch_hessm_f32_##nb(kd, a, grid, s);
    GP_FOR_EACH_KERNEL_NB(GP_CASE)
#undef GP_CASE
  }
  return hipErrorInvalidValue;
}
template <>
hipError_t launch_hessm<double>(int knb, int kd, const gpk::HessMfmaArgs<double>& a, int grid, hipStream_t s) {
  switch (knb) {
#define GP_CASE(nb) case nb: return gpk::launch_hessm_f64_##nb(kd, a, grid, s);
    GP_FOR_EACH_KERNEL_NB(GP_CASE)
#undef GP_CASE
  }
  return hipErrorInvalidValue;
}

// The matrix-core Hessian runs the windowed kernel (gp_hessian_win_kernel.hpp) or, with GP_HESS_WIN=0,
// the block-major hessian_mfma_kernel (A/B reference).  Read once: a model's packed products follow
// the kernel that will read them.
static bool hess_use_win() {
  static const bool v = [] { const char* ev = getenv("GP_HESS_WIN"); return !ev || atoi(ev) != 0; }();
  return v;
}

static bool hessian_on_matrix_core(const gp_model* m) {
  if (m->kernel_nb <= 0 || m->n_emulators != 1 || m->xs_host.empty()) return false;
  if (const char* ev = getenv("GP_HESS_VALU"))      // A/B switch: force the VALU kernel
    if (atoi(ev) != 0) return false;
  return m->kernel_d == 8 || m->kernel_d == 10 || m->kernel_d == 11 || m->kernel_d == 12 || m->kernel_d == 16;
}

// The constant operand of hessian_mfma_kernel, built once per model: P[i][(d, d2)] =
// x''_id x''_id2 (double product of the rounded coordinates, rounded once to T) in 4 x 4 blocks
// of (d, d2); fragment (block c, training block I, k-step s) lane l = the product for training
// point 16 I + own(s, l >> 4) and the block's element that MFMA output row (l & 15) stands for.
template <typename T>
static int ensure_hess_frags(gp_ctx* ctx, gp_model* m) {
  std::lock_guard<std::mutex> lock(m->h_mutex);
  if (m->d_pfrags) return GP_OK;
  const int kd = m->kernel_d, knb = m->kernel_nb, N = m->n_train;
  // geometry of hessian_mfma_kernel<T, kd, knb> (gpk::HGeo): wide instances take their
  // fragments in paired order and in chunks of one pair block
  const bool win = gpk::hess_win<T>(kd, knb) && hess_use_win();   // windowed kernel: k-step-major fragments, 32 per chunk
  const bool wide = !win && gpk::hess_wide<T>(kd, knb);
  const int chunk = win ? gpk::WGeo::kChunk : wide ? 4 * knb : gpk::Geo<T>::kChunk;
  const int nblk = gpk::hess_blocks(kd);
  const size_t n = (size_t)gpk::hess_frag_count_padded(kd, knb, chunk) * 64;
  std::vector<T> fr(n, T(0));
  for (int c = 0; c < nblk; ++c)
    for (int I = 0; I < knb; ++I)
      for (int s = 0; s < 4; ++s) {
        T* f = fr.data() + (size_t)(win ? gpk::hess_win_frag_index(c, I, s, nblk)
                                        : gpk::hess_frag_index(c, I, s, knb, wide, nblk)) * 64;
        for (int l = 0; l < 64; ++l) {
          const int i = gpk::own_index<T>(I, s, l >> 4);   // slot; padding slots hold zero rows
          const int q = l & 15;             // MFMA output row = accumulator r of lane group g
          const int d = 4 * gpk::hess_block_bi(c) + gpk::hess_row_r<T>(q);
          const int d2 = 4 * gpk::hess_block_bj(c) + gpk::hess_row_g<T>(q);
          if (gpk::slot_point<T>(i) >= N) continue;
          if (win && gpk::hess_block_bi(c) == gpk::hess_block_bj(c)) {
            // the windowed kernel takes G_n = sum w x''_n and s = sum w from the unused mirror slots of the
            // diagonal blocks (gp_hessian_win_kernel.hpp, hess_gslot_*)
            const int n = gpk::hess_gslot_of(gpk::hess_block_bi(c), gpk::hess_row_r<T>(q), gpk::hess_row_g<T>(q));
            if (n >= 0) {
              f[l] = n < kd ? (T)m->xs_host[(size_t)i * kd + n] : n == kd ? T(1) : T(0);
              continue;
            }
          }
          if (d >= kd || d2 >= kd) continue;
          f[l] = (T)(m->xs_host[(size_t)i * kd + d] * m->xs_host[(size_t)i * kd + d2]);
        }
      }
  void* dp = nullptr;
  HIP_TRY(hipMalloc(&dp, sizeof(T) * n));
  hipError_t e = hipMemcpy(dp, fr.data(), sizeof(T) * n, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(dp);
    return fail(GP_ERR_HIP, "hessian operand upload: %s", hipGetErrorString(e));
  }
  m->d_pfrags = dp;
  return GP_OK;
}

template <typename T>
static int hessian_device(gp_ctx* ctx, const gp_model* m, const void* d_testing, void* d_hess, int64_t M,
                          hipStream_t stream = nullptr) {
  if (!stream) stream = ctx->stream;
  if (m->n_inputs > GP_MAX_KERNEL_D)
    return fail(GP_ERR_UNSUPPORTED, "hessian kernels are compiled for n_inputs <= %d", GP_MAX_KERNEL_D);
  if (hessian_on_matrix_core(m)) {
    int rc = ensure_hess_frags<T>(ctx, const_cast<gp_model*>(m));
    if (rc) return rc;
    gpk::HessMfmaArgs<T> h;
    h.xa = (const T*)m->d_xa;
    h.pfrags = (const T*)m->d_pfrags;
    h.sd = (const T*)m->d_sd;
    h.testing = (const T*)d_testing;
    h.hess = (T*)d_hess;
    h.M = M;
    h.d_actual = m->n_inputs;
    h.use_win = gpk::hess_win<T>(m->kernel_d, m->kernel_nb) && hess_use_win() ? 1 : 0;
    const bool wide = !h.use_win && gpk::hess_wide<T>(m->kernel_d, m->kernel_nb);
    h.dbg = (unsigned long long*)ctx->dbg;
    h.n_ksteps = (m->n_train + 3) / 4;
    // items drawn from a counter (see the kernel); GP_HESS_STATIC=1: dealt round-robin as in round 2 (A/B)
    static const bool static_items = [] { const char* ev = getenv("GP_HESS_STATIC"); return ev && atoi(ev) != 0; }();
    h.tickets = h.tickets2 = nullptr;
    if (h.use_win && !static_items && M < ((int64_t)1 << 36)) {      // (two launches per call at most: whole groups, rest)
      const unsigned slot = ctx->ticket_next.fetch_add(2);
      h.tickets = ctx->tickets + 4 * (slot % kTicketSlots);            // (4 words per launch: see the kernel)
      h.tickets2 = ctx->tickets + 4 * ((slot + 1) % kTicketSlots);
    }
    const int kRowsPerWG = (wide || h.use_win) ? 4 * gpk::kTile : gpk::Geo<T>::kRowsPerWG;   // 4-wave workgroups
    const int64_t groups = (M + kRowsPerWG - 1) / kRowsPerWG;
    int64_t grid = (int64_t)ctx->compute_units * (h.use_win ? gpk::win_wg_per_cu<T>() : gpk::Geo<T>::kWGPerCU);
    if (grid > groups && !h.use_win) grid = groups;      // (the windowed kernel's launcher sizes its own launches)
    hipError_t e = launch_hessm<T>(m->kernel_nb, m->kernel_d, h, (int)grid, stream);
    if (e != hipSuccess) return fail(GP_ERR_HIP, "hessian kernel launch: %s", hipGetErrorString(e));
    return GP_OK;
  }
  gpk::HessianArgs<T> a;
  a.xa = (const T*)m->d_xa;
  a.sd = (const T*)m->d_sd;
  a.testing = (const T*)d_testing;
  a.hess = (T*)d_hess;
  a.M = M;
  a.d_actual = m->n_inputs;
  a.nb = (m->n_train + 15) / 16;     // the loop over training points is a run-time loop
  if (sizeof(T) * (16 * (size_t)a.nb * gpk::row_stride(m->kernel_d) + 2 * m->kernel_d) > 160 * 1024)
    return fail(GP_ERR_UNSUPPORTED, "training set too large for the hessian kernel's LDS image");
  const int64_t groups = (M + gpk::hkRowsPerWG - 1) / gpk::hkRowsPerWG;
  int64_t grid = (int64_t)ctx->compute_units * 2;
  if (grid > groups) grid = groups;
  hipError_t e = launch_hessian<T>(m->kernel_d, a, (int)grid, stream);
  if (e != hipSuccess) return fail(GP_ERR_HIP, "hessian kernel launch: %s", hipGetErrorString(e));
  return GP_OK;
}

static int ensure_scratch(gp_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return GP_OK;
  if (ctx->scratch) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
  }
  HIP_TRY(hipMalloc(&ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return GP_OK;
}

// ---- host-pointer path --------------------------------------------------------------------
// Host arrays in, host arrays out (what every caller of the reference's predict() has).  The
// rows are cut into slabs that flow through kPipeSlots slots; slot k owns a stream, pinned
// staging buffers and device buffers.  For slab s the calling thread and the context's helper
// threads (gp_host_pool.hpp)
//   1. wait for slab s - kPipeSlots (same slot) to come back and copy it out of pinned staging
//      into the caller's arrays,
//   2. copy slab s from the caller's rows into pinned staging (converting the caller's type TH
//      to the compute type T on the way: a float32 predict on float64 numpy arrays needs no
//      numpy casts at all),
//   3. enqueue H2D, kernel, D2H and an event on the slot's stream,
// so the device works on up to kPipeSlots slabs while the host copies.  Nothing is allocated,
// spawned or uploaded per call once the context is warm.  Measured on the GPU box
// (profiles/r02_host_path_experiments.txt): 1e6 float64 rows of N=250, D=11 take 3.1 ms against
// 9.6 ms in round 1; the host copies alone would take 1.1 ms, the kernels 1.4 ms, and H2D + D2H
// of the 192 MB 3.0-3.5 ms -- the two directions do not overlap on this host, whatever streams
// issue them -- so the path sits on its PCIe floor.  Tried and dropped: more slots (slower from
// 4 up), non-temporal host copies (no gain), letting the kernel read / write pinned host memory
// itself instead of H2D / D2H copies (4.1 ms), one stream per copy direction (no change),
// pre-touching or huge-page advice for fresh output arrays (no gain; see _lib.OutputPool).
// GP_HOST_TRACE=1 prints where a call's time went (event waits / host copies / enqueue).
// NUMA node the device hangs off (sysfs, by PCI bus id); -1 when it cannot be told.
static int device_numa_node(int device) {
  char bdf[64] = "";
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return -1;
  for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE* fh = fopen(path, "r");
  if (!fh) return -1;
  int node = -1;
  if (fscanf(fh, "%d", &node) != 1) node = -1;
  fclose(fh);
  return node;
}
// the cpus of that node that this process may run on (empty set: unknown / none)
static bool node_cpus(int node, cpu_set_t* out) {
  CPU_ZERO(out);
  if (node < 0) return false;
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* fh = fopen(path, "r");
  if (!fh) return false;
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) { fclose(fh); return false; }
  int a, b, n = 0;
  while (fscanf(fh, "%d", &a) == 1) {
    b = a;
    int ch = fgetc(fh);
    if (ch == '-') { if (fscanf(fh, "%d", &b) != 1) break; ch = fgetc(fh); }
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &allowed)) { CPU_SET(c, out); ++n; }
    if (ch != ',') break;
  }
  fclose(fh);
  return n > 0;
}

static int host_threads() {
  static int n = [] {
    int avail = 1;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) avail = CPU_COUNT(&set);
    const char* ev = getenv("GP_HOST_THREADS");
    int v = ev ? atoi(ev) : 8;
    if (v > avail) v = avail;
    return v < 1 ? 1 : (v > 32 ? 32 : v);
  }();
  return n;
}

// The helper threads run on the cpus of the device's NUMA node (when sysfs tells and the process
// may use them; GP_HOST_PIN=0 turns it off): the pinned staging buffers they copy into and out of
// are then allocated and touched next to the PCIe root the DMA goes through.
static gph::ThreadPool& host_pool(gp_ctx* ctx) {
  if (!ctx->pipe.pool) {
    cpu_set_t cpus;
    const char* ev = getenv("GP_HOST_PIN");
    const bool pin = (!ev || atoi(ev) != 0) && node_cpus(device_numa_node(ctx->device), &cpus);
    ctx->pipe.pool.reset(new gph::ThreadPool(host_threads(), pin ? &cpus : nullptr));
  }
  return *ctx->pipe.pool;
}


template <typename TD, typename TS>
static inline void convert_range(TD* dst, const TS* src, size_t lo, size_t hi) {
  if (sizeof(TD) == sizeof(TS)) std::memcpy((void*)(dst + lo), (const void*)(src + lo), (hi - lo) * sizeof(TD));
  else for (size_t i = lo; i < hi; ++i) dst[i] = (TD)src[i];
}

// (dev_only_bytes > 0: the slots' device buffers alone -- the pinned-array path stages nothing)
static int ensure_pipe(gp_ctx* ctx, size_t in_bytes, size_t out_bytes, size_t dev_only_bytes = 0) {
  gp_pipe& pp = ctx->pipe;
  if (!pp.ready) {
    for (int k = 0; k < kPipeSlots; ++k) {
      if (!pp.stream[k]) HIP_TRY(hipStreamCreateWithFlags(&pp.stream[k], hipStreamNonBlocking));
      if (!pp.done[k]) HIP_TRY(hipEventCreateWithFlags(&pp.done[k], hipEventDisableTiming));
      if (!pp.in_there[k]) HIP_TRY(hipEventCreateWithFlags(&pp.in_there[k], hipEventDisableTiming));
      if (!pp.computed[k]) HIP_TRY(hipEventCreateWithFlags(&pp.computed[k], hipEventDisableTiming));
    }
    if (!pp.up) HIP_TRY(hipStreamCreateWithFlags(&pp.up, hipStreamNonBlocking));
    if (!pp.down) HIP_TRY(hipStreamCreateWithFlags(&pp.down, hipStreamNonBlocking));
    pp.ready = true;
  }
  // grow-only; a size is recorded only after every slot's buffer exists, so a failed
  // allocation can never leave a stale size beside a null buffer
  // (pinned staging is allocated by a helper thread, i.e. on the device's NUMA node when the
  // helpers are pinned there -- the calling thread may sit on the other socket)
  auto host_alloc = [&](void** slot, size_t bytes) -> hipError_t {
    hipError_t err = hipSuccess;
    const int dev = ctx->device;
    host_pool(ctx).run_on_worker([&] {
      err = hipSetDevice(dev);
      if (err == hipSuccess) err = hipHostMalloc(slot, bytes, hipHostMallocDefault);
      if (err == hipSuccess) std::memset(*slot, 0, bytes);       // first touch
    });
    return err;
  };
  if (dev_only_bytes == 0 && pp.stage_in_bytes < in_bytes) {
    pp.stage_in_bytes = 0;
    for (int k = 0; k < kPipeSlots; ++k) {
      if (pp.stage_in[k]) { void* q = pp.stage_in[k]; pp.stage_in[k] = nullptr; HIP_TRY(hipHostFree(q)); }
      HIP_TRY(host_alloc(&pp.stage_in[k], in_bytes));
    }
    pp.stage_in_bytes = in_bytes;
  }
  if (dev_only_bytes == 0 && pp.stage_out_bytes < out_bytes) {
    pp.stage_out_bytes = 0;
    for (int k = 0; k < kPipeSlots; ++k) {
      if (pp.stage_out[k]) { void* q = pp.stage_out[k]; pp.stage_out[k] = nullptr; HIP_TRY(hipHostFree(q)); }
      HIP_TRY(host_alloc(&pp.stage_out[k], out_bytes));
    }
    pp.stage_out_bytes = out_bytes;
  }
  const size_t dev_need = dev_only_bytes ? dev_only_bytes : in_bytes + out_bytes;
  if (pp.dev_bytes < dev_need) {
    pp.dev_bytes = 0;
    for (int k = 0; k < kPipeSlots; ++k) {
      if (pp.dev[k]) { void* q = pp.dev[k]; pp.dev[k] = nullptr; HIP_TRY(hipFree(q)); }
      HIP_TRY(hipMalloc(&pp.dev[k], dev_need));
    }
    pp.dev_bytes = dev_need;
  }
  return GP_OK;
}

// Rows per slab.  One emulator: two rounds of the persistent grid (65 536 rows in fp64), small
// enough that the first copy-in and the last copy-out -- the only parts nothing overlaps --
// stay short, large enough that launches and thread hand-offs do not show; calls that would be
// one or two slabs are cut into four so that something overlaps.  Batched emulators: as many
// rows as keep a slot's output staging near 16 MiB.  max_rows (> 0) bounds it from above:
// gpu_predict's `threshold` (no more than that many rows are on the device per launch).
template <typename T>
static int64_t slab_rows(const gp_ctx* ctx, const gp_model* m, int64_t M, int64_t out_row_elems, int64_t max_rows) {
  constexpr int64_t kRowsPerWG = gpk::Geo<T>::kRowsPerWG;
  static const int rounds = [] { const char* ev = getenv("GP_HOST_SLAB_ROUNDS"); const int v = ev ? atoi(ev) : 2; return v < 1 ? 1 : v; }();
  int64_t slab = rounds * (int64_t)ctx->compute_units * gpk::Geo<T>::kWGPerCU * kRowsPerWG;
  // (batched emulators: 64 MiB of results per slab, so that a slab is a few hundred rows and an
  // emulator's share of it a copy of kilobytes, not of bytes)
  const int64_t by_bytes = ((int64_t)(m->n_emulators > 1 ? 64 : 16) << 20) / (int64_t)(out_row_elems * sizeof(T));
  if (slab > by_bytes) slab = by_bytes;
  if (M < 4 * slab && M >= 4 * 8192) slab = (M + 3) / 4;
  slab = (slab + kRowsPerWG - 1) / kRowsPerWG * kRowsPerWG;
  if (max_rows > 0 && slab > max_rows) slab = max_rows;
  if (slab > M) slab = M;
  return slab < 1 ? 1 : slab;
}

// The slab pipeline.  in_row / out_row: elements per row in the staging buffers (compute type T).
//   copy_in(stage, s0, n, lo, hi)   rows [lo, hi) of slab [s0, s0 + n): caller's rows -> stage
//   launch(d_in, d_out, n, stream)  enqueue the kernel(s) for a slab
//   copy_out(stage, s0, n, lo, hi)  rows [lo, hi) of the slab: stage -> caller's arrays
template <typename T, typename FIn, typename FLaunch, typename FOut>
static int run_slab_pipeline(gp_ctx* ctx, int64_t M, int64_t slab, size_t in_row, size_t out_row,
                             FIn copy_in, FLaunch launch, FOut copy_out) {
  const size_t in_elems = (size_t)slab * in_row, out_elems = (size_t)slab * out_row;
  int rc = ensure_pipe(ctx, in_elems * sizeof(T), out_elems * sizeof(T));
  if (rc) return rc;
  gp_pipe& pp = ctx->pipe;
  const int64_t ns = (M + slab - 1) / slab;
  auto rows_of = [&](int64_t s) { return (s + 1) * slab <= M ? slab : M - s * slab; };
  // split a slab's rows into tasks of >= 256 KiB each, at most 2 per thread
  auto tasks_for = [&](int64_t n, size_t row_bytes) {
    const size_t bytes = (size_t)n * row_bytes;
    int t = (int)(bytes / ((size_t)256 << 10));
    const int cap = 2 * host_threads();
    return t < 1 ? 1 : (t > cap ? cap : t);
  };
  static const bool trace = [] { const char* ev = getenv("GP_HOST_TRACE"); return ev && atoi(ev) != 0; }();
  auto now = [] { return std::chrono::steady_clock::now(); };
  double t_wait = 0, t_copy = 0, t_enq = 0;
  const auto t_begin = now();
  hipError_t e = hipSuccess;
  for (int64_t s = 0; s < ns + kPipeSlots && e == hipSuccess && rc == GP_OK; ++s) {
    const int k = (int)(s % kPipeSlots);
    const int64_t so = s - kPipeSlots;            // slab whose results come back now
    const bool has_out = so >= 0 && so < ns, has_in = s < ns;
    if (!has_out && !has_in) continue;
    const int64_t n_out = has_out ? rows_of(so) : 0, n_in = has_in ? rows_of(s) : 0;
    auto t0 = now();
    if (has_out) {
      e = hipEventSynchronize(pp.done[k]);        // slab so: D2H finished, slot k is free
      if (e != hipSuccess) break;
    }
    auto t1 = now();
    const int t_out = has_out ? tasks_for(n_out, out_row * sizeof(T)) : 0;
    const int t_in = has_in ? tasks_for(n_in, in_row * sizeof(T)) : 0;
    const T* o_stage = (const T*)pp.stage_out[k];
    T* i_stage = (T*)pp.stage_in[k];
    // copy-in tasks first (the device waits for them), then copy-out
    auto task = [&](int t) {
      if (t < t_in) {
        const int64_t lo = n_in * t / t_in, hi = n_in * (t + 1) / t_in;
        copy_in(i_stage, s * slab, n_in, lo, hi);
      } else {
        const int u = t - t_in;
        const int64_t lo = n_out * u / t_out, hi = n_out * (u + 1) / t_out;
        copy_out(o_stage, so * slab, n_out, lo, hi);
      }
    };
    const int n_tasks = t_in + t_out;
    static const int skip_host = [] { const char* ev = getenv("GP_HOST_SKIP"); return ev ? atoi(ev) & 8 : 0; }();   // (timing diagnostics only)
    if (skip_host) {}
    else if (n_tasks == 1) task(0);
    else host_pool(ctx).run(n_tasks, task);
    auto t2 = now();
    if (has_in) {
      // GP_HOST_SKIP (timing diagnostics only, results are wrong): bit 0 skips the H2D copies,
      // bit 1 the kernel, bit 2 the D2H copies
      static const int skip = [] { const char* ev = getenv("GP_HOST_SKIP"); return ev ? atoi(ev) : 0; }();
      T* d_in = (T*)pp.dev[k];
      T* d_out = d_in + in_elems;
      // The link is full duplex (tools/pcie_duplex.hip: 57 GB/s one way, 2 x 48 GB/s both ways at once) -- but only
      // for copies that sit in DIFFERENT queues: an upload and a download issued on one stream run one after the
      // other (26 + 26 GB/s), and with a slab's upload, kernel and download all on its slot's stream the download
      // of slab s and the upload of slab s + 1 took turns.  So all uploads go on one stream, all downloads on
      // another, and the slot's stream carries the kernel between two events.
      // (GP_PIPE_DIRSTREAMS: bit 0 = uploads on their own stream, bit 1 = downloads on their own stream; default 3)
      static const int dir_streams = [] { const char* ev = getenv("GP_PIPE_DIRSTREAMS"); return ev ? atoi(ev) : 3; }();
      const bool own_up = dir_streams & 1, own_down = dir_streams & 2;
      hipStream_t s_up = own_up ? pp.up : pp.stream[k], s_down = own_down ? pp.down : pp.stream[k];
      if (!(skip & 1))
        e = hipMemcpyAsync(d_in, pp.stage_in[k], sizeof(T) * (size_t)n_in * in_row, hipMemcpyHostToDevice, s_up);
      if (e != hipSuccess) break;
      if (own_up) {
        e = hipEventRecord(pp.in_there[k], s_up);
        if (e == hipSuccess) e = hipStreamWaitEvent(pp.stream[k], pp.in_there[k], 0);
        if (e != hipSuccess) break;
      }
      if (!(skip & 2)) rc = launch(d_in, d_out, n_in, pp.stream[k]);
      if (rc) break;
      if (own_down) {
        e = hipEventRecord(pp.computed[k], pp.stream[k]);
        if (e == hipSuccess) e = hipStreamWaitEvent(s_down, pp.computed[k], 0);
        if (e != hipSuccess) break;
      }
      if (!(skip & 4))
        e = hipMemcpyAsync(pp.stage_out[k], d_out, sizeof(T) * (size_t)n_in * out_row, hipMemcpyDeviceToHost, s_down);
      if (e == hipSuccess) e = hipEventRecord(pp.done[k], s_down);
    }
    if (trace) {
      auto t3 = now();
      t_wait += std::chrono::duration<double>(t1 - t0).count();
      t_copy += std::chrono::duration<double>(t2 - t1).count();
      t_enq += std::chrono::duration<double>(t3 - t2).count();
    }
  }
  // leave every stream idle whatever happened (the buffers are reused by the next call)
  for (int k = 0; k < kPipeSlots; ++k) (void)hipStreamSynchronize(pp.stream[k]);
  (void)hipStreamSynchronize(pp.up);
  (void)hipStreamSynchronize(pp.down);
  if (trace)
    fprintf(stderr, "[gp host pipeline] rows=%lld slab=%lld slabs=%lld threads=%d: total %.3f ms = event waits %.3f + host copies %.3f + enqueue %.3f\n",
            (long long)M, (long long)slab, (long long)ns, host_threads(),
            std::chrono::duration<double>(now() - t_begin).count() * 1e3, t_wait * 1e3, t_copy * 1e3, t_enq * 1e3);
  if (rc) return rc;
  if (e != hipSuccess) return fail(GP_ERR_HIP, "host pipeline: %s", hipGetErrorString(e));
  return GP_OK;
}

// Is [p, p + bytes) page-locked host memory the device can copy to / from directly (gp_pinned_alloc,
// hipHostMalloc, hipHostRegister)?  Pageable memory makes hipPointerGetAttributes fail: not an error here.
static bool is_pinned_host(const void* p, size_t bytes) {
  if (!p) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (a.type != hipMemoryTypeHost) return false;
  hipPointerAttribute_t b;      // (the last byte too: a registered range may end inside the array)
  if (hipPointerGetAttributes(&b, (const char*)p + (bytes ? bytes - 1 : 0)) != hipSuccess) { (void)hipGetLastError(); return false; }
  return b.type == hipMemoryTypeHost;
}

// The host-pointer path for a caller whose rows AND result arrays are page-locked (gp_pinned_alloc): no staging
// and no host copies at all -- every slab's rows go from the caller's array to the device and its results straight
// into the caller's arrays (row-major gradient, one emulator), uploads and downloads on their own streams so that
// the two directions of the link run at once.  What the slab pipeline spends on 384 MB of host copies per 1e6 rows
// (the main thread's critical path, tools/r03_host_matrix.sh) is gone, and with it the host-DRAM traffic that
// limits several GPUs streaming at once.
template <typename T>
static int predict_host_pinned(gp_ctx* ctx, const gp_model* m, const T* testing, T* result, T* error, T* deriv,
                               int64_t M, int64_t max_rows) {
  const int D = m->n_inputs;
  // Slabs of four rounds of the persistent grid (131 072 rows in fp64) behind a short run-up of one and two rounds
  // (so that the first results are on their way down after ~0.1 ms).  A slab costs one upload and three downloads
  // (mean, variance, gradient: three arrays of the caller's); between two copies on one stream the copy engine
  // idles ~10 us, and copies of less than a few MB do not reach the link's rate anyway (tools/pcie_duplex.hip:
  // 1 MB pieces 33 GB/s per direction, 6 MB pieces 45).  Measured: slabs doubling up to eight rounds 2.83 ms per
  // 1e6 rows (while the slabs grow, a slab's results are down before the next, twice as long, upload and kernel
  // are through), the two small downloads on a second download stream 3.30 ms (it contends with the first).
  // max_rows bounds the slabs.
  constexpr int64_t kRowsPerWG = gpk::Geo<T>::kRowsPerWG;
  const int64_t round = (int64_t)ctx->compute_units * gpk::Geo<T>::kWGPerCU * kRowsPerWG;
  int64_t cap = 4 * round;
  if (max_rows > 0 && cap > max_rows) cap = max_rows < kRowsPerWG ? kRowsPerWG : max_rows / kRowsPerWG * kRowsPerWG;
  const size_t in_elems = (size_t)cap * D, out_elems = (size_t)cap * (2 + D);
  int rc = ensure_pipe(ctx, 0, 0, (in_elems + out_elems) * sizeof(T));
  if (rc) return rc;
  gp_pipe& pp = ctx->pipe;
  hipError_t e = hipSuccess;
  int64_t s0 = 0, cur = round < cap ? round : cap;
  for (int64_t s = 0; s0 < M && e == hipSuccess && rc == GP_OK; ++s) {
    const int k = (int)(s % kPipeSlots);
    const int64_t n = s0 + cur <= M ? cur : M - s0;
    if (s >= kPipeSlots) e = hipEventSynchronize(pp.done[k]);       // the slot's device buffers are free again
    if (e != hipSuccess) break;
    T* d_in = (T*)pp.dev[k];
    T* d_mu = d_in + in_elems;
    T* d_var = d_mu + n;
    T* d_der = d_var + n;
    e = hipMemcpyAsync(d_in, testing + (size_t)s0 * D, sizeof(T) * (size_t)n * D, hipMemcpyHostToDevice, pp.up);
    if (e == hipSuccess) e = hipEventRecord(pp.in_there[k], pp.up);
    if (e == hipSuccess) e = hipStreamWaitEvent(pp.stream[k], pp.in_there[k], 0);
    if (e != hipSuccess) break;
    rc = predict_device<T>(ctx, m, d_in, d_mu, d_var, d_der, n, GP_DERIV_ROWMAJOR, pp.stream[k]);
    if (rc) break;
    e = hipEventRecord(pp.computed[k], pp.stream[k]);
    if (e == hipSuccess) e = hipStreamWaitEvent(pp.down, pp.computed[k], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(deriv + (size_t)s0 * D, d_der, sizeof(T) * (size_t)n * D, hipMemcpyDeviceToHost, pp.down);
    if (e == hipSuccess) e = hipMemcpyAsync(result + s0, d_mu, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost, pp.down);
    if (e == hipSuccess) e = hipMemcpyAsync(error + s0, d_var, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost, pp.down);
    if (e == hipSuccess) e = hipEventRecord(pp.done[k], pp.down);
    s0 += n;
    if (2 * cur <= cap) cur *= 2;
  }
  for (int k = 0; k < kPipeSlots; ++k) (void)hipStreamSynchronize(pp.stream[k]);
  (void)hipStreamSynchronize(pp.up);
  (void)hipStreamSynchronize(pp.down);
  if (rc) return rc;
  if (e != hipSuccess) return fail(GP_ERR_HIP, "host pipeline (pinned arrays): %s", hipGetErrorString(e));
  return GP_OK;
}

// predict for host arrays.  T = compute type (the model's), TH = the caller's host type (T, or
// double with T = float).  Outputs: result/error [E][M], deriv [E][M*D] (row-major) or [E][D][M].
template <typename T, typename TH>
static int predict_host(gp_ctx* ctx, const gp_model* m, const TH* testing, TH* result, TH* error,
                        TH* deriv, int64_t M, int layout, int64_t max_rows) {
  const int D = m->n_inputs, E = m->n_emulators;
  if (M == 0) return GP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t out_row = (size_t)E * (2 + D);
  // Small calls (one slab, no type conversion): no staging, no helper threads -- the rows go
  // straight from the caller's array to the device and the results straight back (the runtime
  // pins the pages for the DMA: 61 us for 2.6 MB on the box, where a staged copy needs a thread
  // hand-off and two memcpys), one launch, one synchronisation.  The device layouts are the
  // caller's ([E][M], [E][M*D] or [E][D][M]) because the slab is the whole call.
  static const int64_t direct_rows = [] { const char* ev = getenv("GP_HOST_DIRECT_ROWS"); return ev ? (int64_t)atoll(ev) : (int64_t)32768; }();
  if (sizeof(T) == sizeof(TH) && M <= direct_rows && (max_rows <= 0 || M <= max_rows) &&
      (size_t)M * out_row * sizeof(T) <= ((size_t)32 << 20)) {
    const size_t n_in = (size_t)M * D, n_e = (size_t)E * M;
    int rc = ensure_scratch(ctx, (n_in + n_e * (2 + D)) * sizeof(T));
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    T* d_in = (T*)ctx->scratch;
    T* d_mu = d_in + n_in;
    T* d_var = d_mu + n_e;
    T* d_der = d_var + n_e;
    HIP_TRY(hipMemcpyAsync(d_in, testing, n_in * sizeof(T), hipMemcpyHostToDevice, st));
    rc = predict_device<T>(ctx, m, d_in, d_mu, d_var, d_der, M, layout, st);
    hipError_t e = hipSuccess;
    if (!rc) {
      // one copy when the caller laid result | error | deriv out back to back
      const bool one = (const void*)error == (const void*)(result + n_e) && (const void*)deriv == (const void*)(error + n_e);
      e = hipMemcpyAsync(result, d_mu, (one ? n_e * (2 + D) : n_e) * sizeof(T), hipMemcpyDeviceToHost, st);
      if (!one && e == hipSuccess) e = hipMemcpyAsync(error, d_var, n_e * sizeof(T), hipMemcpyDeviceToHost, st);
      if (!one && e == hipSuccess) e = hipMemcpyAsync(deriv, d_der, n_e * D * sizeof(T), hipMemcpyDeviceToHost, st);
    }
    const hipError_t es = hipStreamSynchronize(st);      // whatever happened, leave the stream idle
    if (rc) return rc;
    if (e != hipSuccess || es != hipSuccess)
      return fail(GP_ERR_HIP, "predict (direct copies): %s", hipGetErrorString(e != hipSuccess ? e : es));
    return GP_OK;
  }
  if constexpr (sizeof(T) == sizeof(TH)) {
    static const bool no_pinned = [] { const char* ev = getenv("GP_NO_PINNED_PATH"); return ev && atoi(ev) != 0; }();
    if (!no_pinned && E == 1 && layout == GP_DERIV_ROWMAJOR &&
        is_pinned_host(testing, (size_t)M * D * sizeof(T)) && is_pinned_host(result, (size_t)M * sizeof(T)) &&
        is_pinned_host(error, (size_t)M * sizeof(T)) && is_pinned_host(deriv, (size_t)M * D * sizeof(T)))
      return predict_host_pinned<T>(ctx, m, (const T*)testing, (T*)result, (T*)error, (T*)deriv, M, max_rows);
  }
  const int64_t slab = slab_rows<T>(ctx, m, M, (int64_t)out_row, max_rows);
  // float64 rows for a float32 model: centre and scale in double, round once (the kernel then
  // takes the rows as they are).  Rounding the raw rows first would cost |t| / |t - c| in
  // relative accuracy of every distance -- 9e-5 of the mean on the PROSAIL emulator.
  const bool prescale = sizeof(T) == 4 && sizeof(TH) == 8 && E == 1 && m->kernel_nb > 0;
  const double* sc = m->scale_host.data();
  const double* ce = m->centre_host.data();
  auto copy_in = [=](T* stage, int64_t s0, int64_t n, int64_t lo, int64_t hi) {
    (void)n;
    const TH* src = testing + (size_t)s0 * D;
    if (prescale) {
      // scale and centre replicated over 8 rows: the loops below run over contiguous memory
      // with a trip count the compiler can vectorise (D itself is 10 or 11)
      double sc8[8 * GP_MAX_KERNEL_D], ce8[8 * GP_MAX_KERNEL_D];
      for (int q = 0; q < 8 * D; ++q) { sc8[q] = sc[q % D]; ce8[q] = ce[q % D]; }
      int64_t r = lo;
      for (; r + 8 <= hi; r += 8) {
        const TH* a = src + (size_t)r * D;
        T* o = stage + (size_t)r * D;
        for (int q = 0; q < 8 * D; ++q) o[q] = (T)(sc8[q] * ((double)a[q] - ce8[q]));
      }
      for (; r < hi; ++r)
        for (int d = 0; d < D; ++d)
          stage[(size_t)r * D + d] = (T)(sc[d] * ((double)src[(size_t)r * D + d] - ce[d]));
    } else {
      convert_range(stage, src, (size_t)lo * D, (size_t)hi * D);
    }
  };
  auto launch = [=](T* d_in, T* d_out, int64_t n, hipStream_t st) {
    return predict_device<T>(ctx, m, d_in, d_out, d_out + (size_t)E * n, d_out + (size_t)2 * E * n, n,
                             layout, st, prescale);
  };
  // staged slab: mu [E][n], var [E][n], deriv [E][n*D] or [E][D][n]
  auto copy_out = [=](const T* o, int64_t s0, int64_t n, int64_t lo, int64_t hi) {
    // a task is a share [lo, hi) of the slab's rows -- or, for batched emulators, the same share of
    // the EMULATORS with all the slab's rows: every emulator's results are a separate run of the
    // caller's arrays, and a row share of each would be a few dozen bytes per copy
    int e_lo = 0, e_hi = E;
    if (E > 1) {
      e_lo = (int)((int64_t)E * lo / n);
      e_hi = (int)((int64_t)E * hi / n);
      lo = 0;
      hi = n;
    }
    for (int e = e_lo; e < e_hi; ++e) {
      convert_range(result + (size_t)e * M + s0, o + (size_t)e * n, (size_t)lo, (size_t)hi);
      convert_range(error + (size_t)e * M + s0, o + (size_t)(E + e) * n, (size_t)lo, (size_t)hi);
      const T* od = o + (size_t)2 * E * n + (size_t)e * n * D;
      TH* hd = deriv + (size_t)e * M * D;
      if (layout == GP_DERIV_ROWMAJOR) {
        convert_range(hd + (size_t)s0 * D, od, (size_t)lo * D, (size_t)hi * D);
      } else {
        for (int d = 0; d < D; ++d)
          convert_range(hd + (size_t)d * M + s0, od + (size_t)d * n, (size_t)lo, (size_t)hi);
      }
    }
  };
  return run_slab_pipeline<T>(ctx, M, slab, (size_t)D, out_row, copy_in, launch, copy_out);
}

// Hessian for host arrays: (M, D, D) out, same pipeline (2 KiB per row of output at D = 16).
template <typename T, typename TH = T>
static int hessian_host_model(gp_ctx* ctx, const gp_model* m, const TH* testing, TH* hess, int64_t M) {
  const int D = m->n_inputs;
  if (M == 0) return GP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t out_row = (size_t)D * D;
  if (sizeof(T) == sizeof(TH) && (size_t)M * out_row * sizeof(T) <= ((size_t)8 << 20)) {      // small call: direct copies (see predict_host)
    const size_t n_in = (size_t)M * D, n_out = (size_t)M * out_row;
    int rc = ensure_scratch(ctx, (n_in + n_out) * sizeof(T));
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    T* d_in = (T*)ctx->scratch;
    T* d_out = d_in + n_in;
    HIP_TRY(hipMemcpyAsync(d_in, testing, n_in * sizeof(T), hipMemcpyHostToDevice, st));
    rc = hessian_device<T>(ctx, m, d_in, d_out, M, st);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(hess, d_out, n_out * sizeof(T), hipMemcpyDeviceToHost, st);
    const hipError_t es = hipStreamSynchronize(st);
    if (rc) return rc;
    if (e != hipSuccess || es != hipSuccess)
      return fail(GP_ERR_HIP, "hessian (direct copies): %s", hipGetErrorString(e != hipSuccess ? e : es));
    return GP_OK;
  }
  const int64_t slab = slab_rows<T>(ctx, m, M, (int64_t)out_row, 0);
  auto copy_in = [=](T* stage, int64_t s0, int64_t, int64_t lo, int64_t hi) {
    convert_range(stage, testing + (size_t)s0 * D, (size_t)lo * D, (size_t)hi * D);
  };
  auto launch = [=](T* d_in, T* d_out, int64_t n, hipStream_t st) {
    return hessian_device<T>(ctx, m, d_in, d_out, n, st);
  };
  auto copy_out = [=](const T* o, int64_t s0, int64_t, int64_t lo, int64_t hi) {
    convert_range(hess + (size_t)s0 * out_row, o, (size_t)lo * out_row, (size_t)hi * out_row);
  };
  return run_slab_pipeline<T>(ctx, M, slab, (size_t)D, out_row, copy_in, launch, copy_out);
}

// ---- the context's model cache (see gp_cached_model) ---------------------------------------
template <typename T, typename TH>
static int cached_model(gp_ctx* ctx, const TH* expX, const TH* inputs, const TH* invQt, const TH* invQ,
                        int N, int D, int theta_size, gp_model** out) {
  *out = nullptr;
  if (!expX || !inputs || !invQt) return fail(GP_ERR_INVALID, "null pointer");
  if (N <= 0 || D <= 0 || theta_size < D + 1) return fail(GP_ERR_INVALID, "bad sizes");
  const int hd = sizeof(TH) == 8 ? GP_F64 : GP_F32, cd = sizeof(T) == 8 ? GP_F64 : GP_F32;
  const size_t b0 = sizeof(TH) * (size_t)theta_size, b1 = sizeof(TH) * (size_t)N * D,
               b2 = sizeof(TH) * (size_t)N, b3 = invQ ? sizeof(TH) * (size_t)N * N : 0;
  gp_cached_model* victim = &ctx->cache[0];
  for (auto& c : ctx->cache) {
    if (c.model && c.host_dtype == hd && c.compute_dtype == cd && c.n_train == N && c.n_inputs == D &&
        c.theta_size == theta_size && c.with_invq == (invQ != nullptr) && c.key.size() == b0 + b1 + b2 + b3) {
      const char* k = c.key.data();
      if (!std::memcmp(k, expX, b0) && !std::memcmp(k + b0, inputs, b1) && !std::memcmp(k + b0 + b1, invQt, b2) &&
          (!invQ || !std::memcmp(k + b0 + b1 + b2, invQ, b3))) {
        c.stamp = ++ctx->cache_clock;
        *out = c.model;
        return GP_OK;
      }
    }
    if (!c.model) { if (victim->model) victim = &c; }
    else if (victim->model && c.stamp < victim->stamp) victim = &c;
  }
  gp_model* m = nullptr;
  int rc = model_create<T, TH>(ctx, 1, expX, inputs, invQt, invQ, N, D, theta_size, &m);
  if (rc) return rc;
  if (victim->model) gp_model_destroy(victim->model);
  victim->model = m;
  victim->host_dtype = hd;
  victim->compute_dtype = cd;
  victim->n_train = N;
  victim->n_inputs = D;
  victim->theta_size = theta_size;
  victim->with_invq = invQ != nullptr;
  victim->key.resize(b0 + b1 + b2 + b3);
  char* k = victim->key.data();
  std::memcpy(k, expX, b0);
  std::memcpy(k + b0, inputs, b1);
  std::memcpy(k + b0 + b1, invQt, b2);
  if (invQ) std::memcpy(k + b0 + b1 + b2, invQ, b3);
  victim->stamp = ++ctx->cache_clock;
  *out = m;
  return GP_OK;
}

// Host-pointer path = predict_wrap: the reference's twelve arguments.
template <typename T, typename TH = T>
static int predict_wrap(gp_ctx* ctx, const TH* expX, const TH* inputs, const TH* invQt,
                        const TH* invQ, const TH* testing, TH* result, TH* error, TH* deriv,
                        int64_t M, int N, int D, int theta_size, int layout = GP_DERIV_DMAJOR) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  if (M < 0) return fail(GP_ERR_INVALID, "n_predict < 0");
  if (M > 0 && (!testing || !result || !error || !deriv)) return fail(GP_ERR_INVALID, "null pointer");
  if (!invQ) return fail(GP_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ctx->device));
  gp_model* m = nullptr;
  int rc = cached_model<T, TH>(ctx, expX, inputs, invQt, invQ, N, D, theta_size, &m);
  if (rc) return rc;
  return predict_host<T, TH>(ctx, m, testing, result, error, deriv, M, layout, 0);
}

// Host-pointer Hessian: constants (cached) + test rows up, (M, D, D) down.
template <typename T>
static int hessian_host(gp_ctx* ctx, const T* expX, const T* inputs, const T* invQt,
                        const T* testing, T* hess, int64_t M, int N, int D, int theta_size) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  if (M < 0) return fail(GP_ERR_INVALID, "n_predict < 0");
  if (M > 0 && (!testing || !hess)) return fail(GP_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ctx->device));
  gp_model* m = nullptr;
  int rc = cached_model<T, T>(ctx, expX, inputs, invQt, (const T*)nullptr, N, D, theta_size, &m);
  if (rc) return rc;
  if (m->n_inputs > GP_MAX_KERNEL_D)
    return fail(GP_ERR_UNSUPPORTED, "hessian kernels are compiled for n_inputs <= %d", GP_MAX_KERNEL_D);
  return hessian_host_model<T>(ctx, m, testing, hess, M);
}

// out[r][band] = sum_p coef[p][r] basis[p][band] on `stream` (gp_reconstruct_kernel.hpp)
static int reconstruct_on(gp_ctx* ctx, int dtype, const void* d_basis, const void* d_coef, void* d_out,
                          int64_t n_rows, int n_pcs, int n_bands, hipStream_t stream) {
  if (n_rows < 0 || n_pcs <= 0 || n_bands <= 0) return fail(GP_ERR_INVALID, "bad sizes");
  if (n_pcs > 16) return fail(GP_ERR_UNSUPPORTED, "reconstruction kernels are compiled for n_pcs <= 16");
  if (n_rows == 0) return GP_OK;
  if (!d_basis || !d_coef || !d_out) return fail(GP_ERR_INVALID, "null device pointer");
  if (dtype != GP_F32 && dtype != GP_F64) return fail(GP_ERR_INVALID, "bad dtype %d", dtype);
  // geometry: rows that one 512-thread workgroup can cover whole (and that a 256-thread one
  // cannot) use the wide form; GP_RECON_WIDE=0/1 overrides for A/B measurements
  const int vec = dtype == GP_F64 ? 2 : 4;
  int wide = (n_bands > 256 * 2 * vec && n_bands <= 512 * 3 * vec) ? 1 : 0;
  if (const char* ev = getenv("GP_RECON_WIDE")) wide = atoi(ev) != 0;
  hipError_t e;
  if (dtype == GP_F64) {
    gpk::ReconArgs<double> a{(const double*)d_basis, (const double*)d_coef, (double*)d_out, n_rows, n_pcs, n_bands};
    e = gpk::launch_reconstruct_f64(a, wide, ctx->compute_units, stream);
  } else {
    gpk::ReconArgs<float> a{(const float*)d_basis, (const float*)d_coef, (float*)d_out, n_rows, n_pcs, n_bands};
    e = gpk::launch_reconstruct_f32(a, wide, ctx->compute_units, stream);
  }
  if (e != hipSuccess) return fail(GP_ERR_HIP, "reconstruct kernel launch: %s", hipGetErrorString(e));
  return GP_OK;
}

// MultivariateEmulator.predict in ONE call (the latency path: an optimiser asks for one state
// vector at a time, multivariate_gp.py:195-222): rows up, the batched predict of all principal
// components, reconstruction and Jacobian on the device, results down, one synchronisation.
// Work buffers come from the context's grow-only device scratch; the copies go straight between
// the caller's arrays and the device (for these sizes the runtime's own pageable path beats a
// staged copy: 61 us against 200 us for 2.6 MB of Jacobians).
constexpr size_t kMvResultMax = (size_t)1 << 30;      // bytes of results per call
// 64-bit digest of host memory blocks (gp_host_digest.cpp: plain host C++, so that it can be compiled with
// per-ISA clones): what a device-resident copy of an emulator was made from
uint64_t gp_host_content_digest(const void* const* blocks, const int64_t* nbytes, int n_blocks);
// The digest of a list of blocks is defined as that of its two halves (the blocks up to the one that takes the
// running length past half of the total, and the rest), combined: the halves can then be taken by two threads.
static inline int digest_split(const int64_t* nbytes, int n_blocks) {
  int64_t total = 0, run = 0;
  for (int b = 0; b < n_blocks; ++b) total += nbytes[b];
  int k = 0;
  while (k < n_blocks && 2 * run < total) run += nbytes[k++];
  return k;
}
static inline uint64_t digest_combine(uint64_t d0, uint64_t d1) {
  return (d0 * 0x9E3779B97F4A7C15ull) ^ ((d1 << 31) | (d1 >> 33));
}
static uint64_t content_digest(const void* const* blocks, const int64_t* nbytes, int n_blocks, gph::ThreadPool* pool = nullptr) {
  const int k = digest_split(nbytes, n_blocks);
  uint64_t d[2];
  auto half = [&](int t) {
    d[t] = t == 0 ? gp_host_content_digest(blocks, nbytes, k) : gp_host_content_digest(blocks + k, nbytes + k, n_blocks - k);
  };
  if (pool) pool->run(2, half);
  else { half(0); half(1); }
  return digest_combine(d[0], d[1]);
}

struct host_check {      // blocks to digest while the device works, and what the digest must be
  const void* const* blocks = nullptr;
  const int64_t* nbytes = nullptr;
  int n_blocks = 0;
  uint64_t expected = 0;
};

template <typename T>
static int mv_predict_host(gp_ctx* ctx, const gp_model* m, const T* d_basis, const T* y, int64_t M,
                           int n_bands, T* fwd, T* jac, const host_check* chk = nullptr) {
  const int D = m->n_inputs, P = m->n_emulators;
  const size_t n_y = (size_t)M * D, n_gp = (size_t)P * M * (2 + D);
  const size_t n_fwd = (size_t)M * n_bands, n_jac = jac ? (size_t)M * D * n_bands : 0;
  if ((n_fwd + n_jac) * sizeof(T) > kMvResultMax)
    return fail(GP_ERR_UNSUPPORTED, "gp_mv_predict_host returns at most %zu MiB per call: split the rows", kMvResultMax >> 20);
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = ensure_scratch(ctx, (n_y + n_gp + n_fwd + n_jac) * sizeof(T));
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  T* d_y = (T*)ctx->scratch;
  T* d_mu = d_y + n_y;
  T* d_var = d_mu + (size_t)P * M;
  T* d_der = d_var + (size_t)P * M;
  T* d_fwd = d_y + n_y + n_gp;
  T* d_jac = d_fwd + n_fwd;
  HIP_TRY(hipMemcpyAsync(d_y, y, n_y * sizeof(T), hipMemcpyHostToDevice, st));
  rc = predict_device<T>(ctx, m, d_y, d_mu, d_var, d_der, M, GP_DERIV_ROWMAJOR, st);
  const int dtype = sizeof(T) == 8 ? GP_F64 : GP_F32;
  if (!rc) rc = reconstruct_on(ctx, dtype, d_basis, d_mu, d_fwd, M, P, n_bands, st);
  if (!rc && jac) rc = reconstruct_on(ctx, dtype, d_basis, d_der, d_jac, M * D, P, n_bands, st);
  // While the device works: is the host data the resident copy was made from still what it was?  Between the
  // launches and the copy back -- a copy into pageable memory does not return before the kernels are through, so
  // behind it there would be nothing left to hide the digest under -- and with hipStreamQuery first: the runtime
  // batches what was enqueued above and would otherwise hand it to the device only when somebody waits.
  bool stale = false;
  if (!rc && chk && chk->n_blocks > 0) {
    (void)hipStreamQuery(st);
    // (two halves, one of them on a helper thread: 45 us alone would outlast the ~35 us the kernels take)
    stale = content_digest(chk->blocks, chk->nbytes, chk->n_blocks, &host_pool(ctx)) != chk->expected;
  }
  hipError_t e = hipSuccess;
  const bool one_copy = jac == fwd + n_fwd;             // the caller laid fwd and jac out back to back
  if (!rc) e = hipMemcpyAsync(fwd, d_fwd, (n_fwd + (one_copy ? n_jac : 0)) * sizeof(T), hipMemcpyDeviceToHost, st);
  if (!rc && e == hipSuccess && jac && !one_copy)
    e = hipMemcpyAsync(jac, d_jac, n_jac * sizeof(T), hipMemcpyDeviceToHost, st);
  const hipError_t es = hipStreamSynchronize(st);       // whatever happened, leave the stream idle
  if (rc) return rc;
  if (e != hipSuccess || es != hipSuccess)
    return fail(GP_ERR_HIP, "mv predict: %s", hipGetErrorString(e != hipSuccess ? e : es));
  return stale ? GP_STALE : GP_OK;
}

extern "C" {

int gp_model_create_f64(gp_ctx* ctx, const double* expX, const double* inputs, const double* invQt,
                        const double* invQ, int n_train, int n_inputs, int theta_size, gp_model** out) {
  return guarded([&] { return model_create<double>(ctx, 1, expX, inputs, invQt, invQ, n_train, n_inputs, theta_size, out); });
}
int gp_batch_create_f64(gp_ctx* ctx, int n_emulators, const double* expX, const double* inputs,
                        const double* invQt, const double* invQ, int n_train, int n_inputs,
                        int theta_size, gp_model** out) {
  return guarded([&] { return model_create<double>(ctx, n_emulators, expX, inputs, invQt, invQ, n_train, n_inputs, theta_size, out); });
}
int gp_batch_create_f32(gp_ctx* ctx, int n_emulators, const float* expX, const float* inputs,
                        const float* invQt, const float* invQ, int n_train, int n_inputs,
                        int theta_size, gp_model** out) {
  return guarded([&] { return model_create<float>(ctx, n_emulators, expX, inputs, invQt, invQ, n_train, n_inputs, theta_size, out); });
}
int gp_model_create_f32_h64(gp_ctx* ctx, const double* expX, const double* inputs, const double* invQt,
                            const double* invQ, int n_train, int n_inputs, int theta_size, gp_model** out) {
  return guarded([&] { return model_create<float, double>(ctx, 1, expX, inputs, invQt, invQ, n_train, n_inputs, theta_size, out); });
}
int gp_batch_create_f32_h64(gp_ctx* ctx, int n_emulators, const double* expX, const double* inputs,
                            const double* invQt, const double* invQ, int n_train, int n_inputs,
                            int theta_size, gp_model** out) {
  return guarded([&] { return model_create<float, double>(ctx, n_emulators, expX, inputs, invQt, invQ, n_train, n_inputs, theta_size, out); });
}
int gp_model_emulators(const gp_model* m, int* n_emulators) {
  if (!m || !n_emulators) return fail(GP_ERR_INVALID, "null pointer");
  *n_emulators = m->n_emulators;
  return GP_OK;
}
int gp_model_create_f32(gp_ctx* ctx, const float* expX, const float* inputs, const float* invQt,
                        const float* invQ, int n_train, int n_inputs, int theta_size, gp_model** out) {
  return guarded([&] { return model_create<float>(ctx, 1, expX, inputs, invQt, invQ, n_train, n_inputs, theta_size, out); });
}

int gp_model_destroy(gp_model* m) {
  if (!m) return GP_OK;
  (void)hipSetDevice(m->device);
  if (m->d_xa) (void)hipFree(m->d_xa);
  if (m->d_frags) (void)hipFree(m->d_frags);
  if (m->d_sd) (void)hipFree(m->d_sd);
  if (m->d_pfrags) (void)hipFree(m->d_pfrags);
  delete m;
  return GP_OK;
}

int gp_model_info(const gp_model* m, int* dtype, int* n_train, int* n_inputs, int* kernel_d, int* kernel_nb) {
  if (!m) return fail(GP_ERR_INVALID, "null model");
  if (dtype) *dtype = m->dtype;
  if (n_train) *n_train = m->n_train;
  if (n_inputs) *n_inputs = m->n_inputs;
  if (kernel_d) *kernel_d = m->kernel_d;
  if (kernel_nb) *kernel_nb = m->kernel_nb;
  return GP_OK;
}

int gp_kernel_ksteps(int n_train, int n_inputs, int* ksteps) {
  if (!ksteps) return fail(GP_ERR_INVALID, "null pointer");
  int kd, knb, knk;
  int rc = pick_kernel(n_train, n_inputs, &kd, &knb, &knk);
  if (rc) return rc;
  *ksteps = knk;
  return GP_OK;
}

int gp_predict_device(gp_ctx* ctx, const gp_model* model, const void* d_testing, void* d_mu,
                      void* d_var, void* d_deriv, int64_t n_predict, int deriv_layout) {
  if (!ctx || !model) return fail(GP_ERR_INVALID, "null context or model");
  if (n_predict < 0) return fail(GP_ERR_INVALID, "n_predict < 0");
  if (n_predict == 0) return GP_OK;
  if (!d_testing || !d_mu || !d_var || !d_deriv) return fail(GP_ERR_INVALID, "null device pointer");
  if (deriv_layout != GP_DERIV_DMAJOR && deriv_layout != GP_DERIV_ROWMAJOR)
    return fail(GP_ERR_INVALID, "bad deriv_layout %d", deriv_layout);
  if (model->device != ctx->device) return fail(GP_ERR_INVALID, "model lives on device %d, context on %d", model->device, ctx->device);
  if (!model->d_frags) return fail(GP_ERR_INVALID, "model was created without invQ: no variance operand");
  HIP_TRY(hipSetDevice(ctx->device));
  if (model->dtype == GP_F64)
    return predict_device<double>(ctx, model, d_testing, d_mu, d_var, d_deriv, n_predict, deriv_layout);
  return predict_device<float>(ctx, model, d_testing, d_mu, d_var, d_deriv, n_predict, deriv_layout);
}

int gp_predict_host(gp_ctx* ctx, const gp_model* model, int host_dtype, const void* testing,
                    void* result, void* error, void* deriv, int64_t n_predict, int deriv_layout,
                    int64_t max_block_rows) {
  if (!ctx || !model) return fail(GP_ERR_INVALID, "null context or model");
  if (n_predict < 0) return fail(GP_ERR_INVALID, "n_predict < 0");
  if (n_predict == 0) return GP_OK;
  if (!testing || !result || !error || !deriv) return fail(GP_ERR_INVALID, "null pointer");
  if (deriv_layout != GP_DERIV_DMAJOR && deriv_layout != GP_DERIV_ROWMAJOR)
    return fail(GP_ERR_INVALID, "bad deriv_layout %d", deriv_layout);
  if (model->device != ctx->device) return fail(GP_ERR_INVALID, "model lives on device %d, context on %d", model->device, ctx->device);
  if (!model->d_frags) return fail(GP_ERR_INVALID, "model was created without invQ: no variance operand");
  if (model->dtype == GP_F64 && host_dtype == GP_F64)
    return guarded([&] { return predict_host<double, double>(ctx, model, (const double*)testing, (double*)result, (double*)error,
                                        (double*)deriv, n_predict, deriv_layout, max_block_rows); });
  if (model->dtype == GP_F32 && host_dtype == GP_F32)
    return guarded([&] { return predict_host<float, float>(ctx, model, (const float*)testing, (float*)result, (float*)error,
                                      (float*)deriv, n_predict, deriv_layout, max_block_rows); });
  if (model->dtype == GP_F32 && host_dtype == GP_F64)
    return guarded([&] { return predict_host<float, double>(ctx, model, (const double*)testing, (double*)result, (double*)error,
                                       (double*)deriv, n_predict, deriv_layout, max_block_rows); });
  return fail(GP_ERR_INVALID, "host arrays must have the model's dtype, or be float64 for a float32 model");
}

int gp_hessian_host(gp_ctx* ctx, const gp_model* model, const void* testing, void* hess, int64_t n_predict) {
  if (!ctx || !model) return fail(GP_ERR_INVALID, "null context or model");
  if (n_predict < 0) return fail(GP_ERR_INVALID, "n_predict < 0");
  if (n_predict == 0) return GP_OK;
  if (!testing || !hess) return fail(GP_ERR_INVALID, "null pointer");
  if (model->device != ctx->device) return fail(GP_ERR_INVALID, "model lives on device %d, context on %d", model->device, ctx->device);
  if (model->n_emulators != 1) return fail(GP_ERR_INVALID, "hessian is per emulator: batch of %d given", model->n_emulators);
  if (model->n_inputs > GP_MAX_KERNEL_D)
    return fail(GP_ERR_UNSUPPORTED, "hessian kernels are compiled for n_inputs <= %d", GP_MAX_KERNEL_D);
  if (model->dtype == GP_F64) return guarded([&] { return hessian_host_model<double>(ctx, model, (const double*)testing, (double*)hess, n_predict); });
  return guarded([&] { return hessian_host_model<float>(ctx, model, (const float*)testing, (float*)hess, n_predict); });
}

int gp_hessian_host_h64(gp_ctx* ctx, const gp_model* model, const double* testing, double* hess, int64_t n_predict) {
  if (!ctx || !model) return fail(GP_ERR_INVALID, "null context or model");
  if (n_predict < 0) return fail(GP_ERR_INVALID, "n_predict < 0");
  if (n_predict == 0) return GP_OK;
  if (!testing || !hess) return fail(GP_ERR_INVALID, "null pointer");
  if (model->device != ctx->device) return fail(GP_ERR_INVALID, "model lives on device %d, context on %d", model->device, ctx->device);
  if (model->n_emulators != 1) return fail(GP_ERR_INVALID, "hessian is per emulator: batch of %d given", model->n_emulators);
  if (model->n_inputs > GP_MAX_KERNEL_D)
    return fail(GP_ERR_UNSUPPORTED, "hessian kernels are compiled for n_inputs <= %d", GP_MAX_KERNEL_D);
  if (model->dtype == GP_F64) return guarded([&] { return hessian_host_model<double>(ctx, model, testing, hess, n_predict); });
  return guarded([&] { return hessian_host_model<float, double>(ctx, model, testing, hess, n_predict); });
}

int gp_device_numa_node(int device, int* node) {
  if (!node) return fail(GP_ERR_INVALID, "null pointer");
  *node = device_numa_node(device);
  return GP_OK;
}

int gp_ctx_host_threads(gp_ctx* ctx, int* n_threads) {
  if (!ctx || !n_threads) return fail(GP_ERR_INVALID, "null pointer");
  *n_threads = host_threads();
  return GP_OK;
}

int gp_predict_wrap_f64(gp_ctx* ctx, const double* expX, const double* inputs, const double* invQt,
                        const double* invQ, const double* testing, double* result, double* error,
                        double* deriv, int64_t n_predict, int n_train, int n_inputs, int theta_size) {
  return guarded([&] { return predict_wrap<double>(ctx, expX, inputs, invQt, invQ, testing, result, error, deriv,
                              n_predict, n_train, n_inputs, theta_size); });
}
int gp_predict_wrap_f32(gp_ctx* ctx, const float* expX, const float* inputs, const float* invQt,
                        const float* invQ, const float* testing, float* result, float* error,
                        float* deriv, int64_t n_predict, int n_train, int n_inputs, int theta_size) {
  return guarded([&] { return predict_wrap<float>(ctx, expX, inputs, invQt, invQ, testing, result, error, deriv,
                             n_predict, n_train, n_inputs, theta_size); });
}

int gp_hessian_device(gp_ctx* ctx, const gp_model* model, const void* d_testing, void* d_hess,
                      int64_t n_predict) {
  if (!ctx || !model) return fail(GP_ERR_INVALID, "null context or model");
  if (n_predict < 0) return fail(GP_ERR_INVALID, "n_predict < 0");
  if (n_predict == 0) return GP_OK;
  if (!d_testing || !d_hess) return fail(GP_ERR_INVALID, "null device pointer");
  if (model->device != ctx->device) return fail(GP_ERR_INVALID, "model lives on device %d, context on %d", model->device, ctx->device);
  if (model->n_emulators != 1) return fail(GP_ERR_INVALID, "hessian is per emulator: batch of %d given", model->n_emulators);
  HIP_TRY(hipSetDevice(ctx->device));
  if (model->dtype == GP_F64) return guarded([&] { return hessian_device<double>(ctx, model, d_testing, d_hess, n_predict); });
  return guarded([&] { return hessian_device<float>(ctx, model, d_testing, d_hess, n_predict); });
}
int gp_hessian_f64(gp_ctx* ctx, const double* expX, const double* inputs, const double* invQt,
                   const double* testing, double* hess, int64_t n_predict, int n_train,
                   int n_inputs, int theta_size) {
  return guarded([&] { return hessian_host<double>(ctx, expX, inputs, invQt, testing, hess, n_predict, n_train, n_inputs, theta_size); });
}
int gp_hessian_f32(gp_ctx* ctx, const float* expX, const float* inputs, const float* invQt,
                   const float* testing, float* hess, int64_t n_predict, int n_train,
                   int n_inputs, int theta_size) {
  return guarded([&] { return hessian_host<float>(ctx, expX, inputs, invQt, testing, hess, n_predict, n_train, n_inputs, theta_size); });
}

int gp_predict_rows_f64(gp_ctx* ctx, const double* expX, const double* inputs, const double* invQt,
                        const double* invQ, const double* testing, double* result, double* error,
                        double* deriv, int64_t n_predict, int n_train, int n_inputs, int theta_size) {
  return guarded([&] { return predict_wrap<double>(ctx, expX, inputs, invQt, invQ, testing, result, error, deriv,
                              n_predict, n_train, n_inputs, theta_size, GP_DERIV_ROWMAJOR); });
}
int gp_predict_rows_f32_h64(gp_ctx* ctx, const double* expX, const double* inputs, const double* invQt,
                            const double* invQ, const double* testing, double* result, double* error,
                            double* deriv, int64_t n_predict, int n_train, int n_inputs, int theta_size) {
  return guarded([&] { return predict_wrap<float, double>(ctx, expX, inputs, invQt, invQ, testing, result, error, deriv,
                                     n_predict, n_train, n_inputs, theta_size, GP_DERIV_ROWMAJOR); });
}
int gp_predict_rows_f32(gp_ctx* ctx, const float* expX, const float* inputs, const float* invQt,
                        const float* invQ, const float* testing, float* result, float* error,
                        float* deriv, int64_t n_predict, int n_train, int n_inputs, int theta_size) {
  return guarded([&] { return predict_wrap<float>(ctx, expX, inputs, invQt, invQ, testing, result, error, deriv,
                             n_predict, n_train, n_inputs, theta_size, GP_DERIV_ROWMAJOR); });
}

int gp_reconstruct_device(gp_ctx* ctx, int dtype, const void* d_basis, const void* d_coef,
                          void* d_out, int64_t n_rows, int n_pcs, int n_bands) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  HIP_TRY(hipSetDevice(ctx->device));
  return reconstruct_on(ctx, dtype, d_basis, d_coef, d_out, n_rows, n_pcs, n_bands, ctx->stream);
}

int gp_mv_predict_host(gp_ctx* ctx, const gp_model* model, const void* d_basis, const void* y,
                       int64_t n_rows, int n_bands, void* fwd, void* jac) {
  if (!ctx || !model) return fail(GP_ERR_INVALID, "null context or model");
  if (n_rows < 0 || n_bands <= 0) return fail(GP_ERR_INVALID, "bad sizes");
  if (n_rows == 0) return GP_OK;
  if (!d_basis || !y || !fwd) return fail(GP_ERR_INVALID, "null pointer");
  if (model->device != ctx->device) return fail(GP_ERR_INVALID, "model lives on device %d, context on %d", model->device, ctx->device);
  if (!model->d_frags) return fail(GP_ERR_INVALID, "model was created without invQ: no variance operand");
  if (model->dtype == GP_F64)
    return guarded([&] { return mv_predict_host<double>(ctx, model, (const double*)d_basis, (const double*)y, n_rows, n_bands, (double*)fwd, (double*)jac); });
  return guarded([&] { return mv_predict_host<float>(ctx, model, (const float*)d_basis, (const float*)y, n_rows, n_bands, (float*)fwd, (float*)jac); });
}

uint64_t gp_content_digest(const void* const* blocks, const int64_t* nbytes, int n_blocks) {
  if (!blocks || !nbytes || n_blocks <= 0) return 0;
  return content_digest(blocks, nbytes, n_blocks);
}

int gp_mv_predict_host_checked(gp_ctx* ctx, const gp_model* model, const void* d_basis, const void* y,
                               int64_t n_rows, int n_bands, void* fwd, void* jac,
                               const void* const* blocks, const int64_t* nbytes, int n_blocks, uint64_t expected) {
  if (!ctx || !model) return fail(GP_ERR_INVALID, "null context or model");
  if (n_rows < 0 || n_bands <= 0 || n_blocks < 0) return fail(GP_ERR_INVALID, "bad sizes");
  if (n_blocks > 0 && (!blocks || !nbytes)) return fail(GP_ERR_INVALID, "null pointer");
  host_check chk;
  chk.blocks = blocks; chk.nbytes = nbytes; chk.n_blocks = n_blocks; chk.expected = expected;
  if (n_rows == 0) return n_blocks > 0 && content_digest(blocks, nbytes, n_blocks) != expected ? GP_STALE : GP_OK;
  if (!d_basis || !y || !fwd) return fail(GP_ERR_INVALID, "null pointer");
  if (model->device != ctx->device) return fail(GP_ERR_INVALID, "model lives on device %d, context on %d", model->device, ctx->device);
  if (!model->d_frags) return fail(GP_ERR_INVALID, "model was created without invQ: no variance operand");
  if (model->dtype == GP_F64)
    return guarded([&] { return mv_predict_host<double>(ctx, model, (const double*)d_basis, (const double*)y, n_rows, n_bands, (double*)fwd, (double*)jac, &chk); });
  return guarded([&] { return mv_predict_host<float>(ctx, model, (const float*)d_basis, (const float*)y, n_rows, n_bands, (float*)fwd, (float*)jac, &chk); });
}

int gp_likelihood_batch_f64(gp_ctx* ctx, int n_sets, const double* theta, const double* inputs,
                            const double* targets, int targets_shared, int n_train, int n_inputs,
                            double* cost, double* grad, double* invQ, double* invQt) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  if (!theta || !inputs || !targets || !cost || !grad) return fail(GP_ERR_INVALID, "null pointer");
  if (n_sets <= 0 || n_train <= 0 || n_inputs <= 0) return fail(GP_ERR_INVALID, "bad sizes");
  if (n_train > gpk::tkMaxN || n_inputs > gpk::tkMaxD)
    return fail(GP_ERR_UNSUPPORTED, "likelihood kernel is compiled for n_train <= %d, n_inputs <= %d",
                gpk::tkMaxN, gpk::tkMaxD);
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t E = (size_t)n_sets, N = (size_t)n_train, D = (size_t)n_inputs;
  const size_t n_theta = E * (D + 2), n_in = N * D, n_tg = (targets_shared ? 1 : E) * N;
  const size_t n_work = E * N * N, n_qt = E * N, n_cost = E, n_grad = E * (D + 2);
  int rc = ensure_scratch(ctx, sizeof(double) * (n_theta + n_in + n_tg + n_work + n_qt + n_cost + n_grad));
  if (rc) return rc;
  double* d_theta = (double*)ctx->scratch;
  double* d_in = d_theta + n_theta;
  double* d_tg = d_in + n_in;
  double* d_work = d_tg + n_tg;
  double* d_qt = d_work + n_work;
  double* d_cost = d_qt + n_qt;
  double* d_grad = d_cost + n_cost;
  hipStream_t st = ctx->stream;
  HIP_TRY(hipMemcpyAsync(d_theta, theta, sizeof(double) * n_theta, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_in, inputs, sizeof(double) * n_in, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_tg, targets, sizeof(double) * n_tg, hipMemcpyHostToDevice, st));
  gpk::TrainArgs a;
  a.theta = d_theta; a.inputs = d_in; a.targets = d_tg;
  a.targets_stride = targets_shared ? 0 : (long long)N;
  a.work = d_work; a.invQt = d_qt; a.cost = d_cost; a.grad = d_grad;
  a.N = n_train; a.D = n_inputs;
  a.full_inverse = invQ != nullptr;
  a.dbg = (unsigned long long*)ctx->dbg;
  hipError_t e = gpk::launch_likelihood(a, n_sets, st);
  if (e != hipSuccess) return fail(GP_ERR_HIP, "likelihood kernel launch: %s", hipGetErrorString(e));
  HIP_TRY(hipMemcpyAsync(cost, d_cost, sizeof(double) * n_cost, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(grad, d_grad, sizeof(double) * n_grad, hipMemcpyDeviceToHost, st));
  if (invQt) HIP_TRY(hipMemcpyAsync(invQt, d_qt, sizeof(double) * n_qt, hipMemcpyDeviceToHost, st));
  if (invQ) HIP_TRY(hipMemcpyAsync(invQ, d_work, sizeof(double) * n_work, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return GP_OK;
}

int gp_pinned_alloc(gp_ctx* ctx, int64_t bytes, void** ptr) {
  if (!ctx || !ptr) return fail(GP_ERR_INVALID, "null pointer");
  if (bytes <= 0) return fail(GP_ERR_INVALID, "bytes must be positive");
  HIP_TRY(hipSetDevice(ctx->device));
  // by a helper thread: they run on the cpus next to the device, so the pages are first touched on its NUMA node
  hipError_t err = hipSuccess;
  const int dev = ctx->device;
  void* q = nullptr;
  host_pool(ctx).run_on_worker([&] {
    err = hipSetDevice(dev);
    if (err == hipSuccess) err = hipHostMalloc(&q, (size_t)bytes, hipHostMallocDefault);
    if (err == hipSuccess) std::memset(q, 0, (size_t)bytes);
  });
  if (err != hipSuccess) return fail(GP_ERR_HIP, "gp_pinned_alloc(%lld bytes): %s", (long long)bytes, hipGetErrorString(err));
  *ptr = q;
  return GP_OK;
}
int gp_pinned_free(gp_ctx* ctx, void* ptr) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  if (!ptr) return GP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipHostFree(ptr));
  return GP_OK;
}
int gp_malloc(gp_ctx* ctx, int64_t bytes, void** dptr) {
  if (!ctx || !dptr) return fail(GP_ERR_INVALID, "null pointer");
  if (bytes <= 0) return fail(GP_ERR_INVALID, "bytes must be positive");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMalloc(dptr, (size_t)bytes));
  return GP_OK;
}
int gp_free(gp_ctx* ctx, void* dptr) {
  if (!ctx) return fail(GP_ERR_INVALID, "null context");
  if (!dptr) return GP_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipFree(dptr));
  return GP_OK;
}
int gp_memcpy_h2d(gp_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  if (!ctx || !dst || !src) return fail(GP_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return GP_OK;
}
int gp_memcpy_d2h(gp_ctx* ctx, void* dst, const void* src, int64_t bytes) {
  if (!ctx || !dst || !src) return fail(GP_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return GP_OK;
}
int gp_memset(gp_ctx* ctx, void* dptr, int value, int64_t bytes) {
  if (!ctx || !dptr) return fail(GP_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemsetAsync(dptr, value, (size_t)bytes, ctx->stream));
  return GP_OK;
}

int gp_event_create(gp_ctx* ctx, gp_event** out) {
  if (!ctx || !out) return fail(GP_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ctx->device));
  gp_event* e = new (std::nothrow) gp_event();
  if (!e) return fail(GP_ERR_NOMEM, "out of host memory");
  e->device = ctx->device;
  hipError_t r = hipEventCreate(&e->ev);
  if (r != hipSuccess) {
    delete e;
    return fail(GP_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(r));
  }
  *out = e;
  return GP_OK;
}
int gp_event_destroy(gp_event* ev) {
  if (!ev) return GP_OK;
  (void)hipSetDevice(ev->device);
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return GP_OK;
}
int gp_event_record(gp_ctx* ctx, gp_event* ev) {
  if (!ctx || !ev) return fail(GP_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipEventRecord(ev->ev, ctx->stream));
  return GP_OK;
}
int gp_event_elapsed_ms(gp_event* start, gp_event* stop, float* ms) {
  if (!start || !stop || !ms) return fail(GP_ERR_INVALID, "null pointer");
  HIP_TRY(hipSetDevice(stop->device));
  HIP_TRY(hipEventSynchronize(stop->ev));
  HIP_TRY(hipEventElapsedTime(ms, start->ev, stop->ev));
  return GP_OK;
}

}  // extern "C"
