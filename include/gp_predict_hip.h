/* gp_predict_hip.h -- C ABI of libgp_predict_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the GP predict hot path of UCL/gp_emulator.  The reference's only
 * native entry point for this path is the CPython-2 extension function
 *
 *     _gpu_predict.predict_wrap(expX, inputs, invQt, invQ, testing,
 *                               result, error, deriv,
 *                               n_predict, n_train, n_inputs, theta_size)
 *
 * (gp_emulator/gpu/_gpu_predict.cpp:115-159, argument parse :86-95, declaration
 * gp_emulator/gpu/gpu_predict.h:46), called from GaussianProcess.gpu_predict
 * (gp_emulator/GaussianProcess.py:313-316).  gp_predict_wrap_f32/_f64 below take exactly
 * those twelve arguments (plus a context handle) with exactly that meaning and layout.
 * Everything else here is the device-resident form of the same path (constants uploaded
 * once, test rows and outputs living in HBM), which the reference does not have because it
 * re-uploads everything per call (gp_emulator/gpu/predict.cu:11-34).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative gp_status; the message for the calling thread's last failure is
 * gp_last_error_string().  Nothing here ever calls exit() (the reference does:
 * _gpu_predict.cpp:45-56, kernel_cdist.cu:28-32, kernel_matrixExp.cu:23-33).
 * All 2-D data is row-major and flattened, as at the reference boundary
 * (GaussianProcess.py:289-291,301).
 */
#ifndef GP_PREDICT_HIP_H
#define GP_PREDICT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gp_status {
  GP_OK = 0,
  GP_STALE = 1,             /* not an error: the host data a device-resident copy was made from has changed
                               (gp_mv_predict_host_checked); discard the results, rebuild the copy, call again */
  GP_ERR_INVALID = -1,      /* bad argument (null pointer, size <= 0, unsupported shape) */
  GP_ERR_HIP = -2,          /* a HIP runtime call failed (see gp_last_error_string) */
  GP_ERR_NO_DEVICE = -3,    /* no usable GPU */
  GP_ERR_UNSUPPORTED = -4,  /* shape outside the compiled kernel set */
  GP_ERR_NOMEM = -5         /* a host allocation failed (no C++ exception ever crosses this boundary) */
} gp_status;

typedef enum gp_dtype { GP_F32 = 0, GP_F64 = 1 } gp_dtype;

/* deriv output layouts */
#define GP_DERIV_DMAJOR 0   /* element (d, m) at d*M + m: the reference boundary's layout
                               (predict.cu:139-150, undone at GaussianProcess.py:321) */
#define GP_DERIV_ROWMAJOR 1 /* (M, D) row-major: what GaussianProcess.predict returns */

typedef struct gp_ctx gp_ctx;     /* one per (thread, device): device id + HIP stream */
typedef struct gp_model gp_model; /* packed per-emulator constants resident in HBM */
typedef struct gp_event gp_event; /* HIP event on the context's stream */

/* ---- library / device ---------------------------------------------------------------- */
const char* gp_last_error_string(void);
const char* gp_version_string(void);
int gp_device_count(int* count);

int gp_ctx_create(int device, gp_ctx** out);
int gp_ctx_destroy(gp_ctx* ctx);
int gp_ctx_synchronize(gp_ctx* ctx);
/* Diagnostic builds only (-DGP_STAMPS=1, tools/stamp_profile.py): device buffer of 8 uint64
 * that the predict kernel adds its per-segment cycle sums to.  Ignored by normal builds. */
int gp_ctx_set_debug_buffer(gp_ctx* ctx, void* d_buffer);
/* number of compute units and HBM bytes of the context's device */
int gp_ctx_device_info(gp_ctx* ctx, int* compute_units, int64_t* hbm_bytes, char* name, int name_len);

/* ---- the reference boundary: predict_wrap --------------------------------------------
 * Replaces _gpu_predict.predict_wrap (gp_emulator/gpu/_gpu_predict.cpp:115-159).
 *   expX      [theta_size]         exp(theta); expX[n_inputs] is the signal variance b
 *   inputs    [n_train*n_inputs]   training inputs, row-major
 *   invQt     [n_train]            invQ . targets
 *   invQ      [n_train*n_train]    row-major; ANY matrix (tests/benchmark.py:14 feeds a
 *                                  random non-symmetric one)
 *   testing   [n_predict*n_inputs] test rows, row-major
 *   result    [n_predict]          out: mean
 *   error     [n_predict]          out: variance  b - k^T invQ k
 *   deriv     [n_inputs*n_predict] out: gradient, DIMENSION-MAJOR (GP_DERIV_DMAJOR)
 * Host pointers in, host pointers out; the call returns when the outputs are written.
 * Differences from the reference, all relaxations: n_predict is 64-bit and has no minimum
 * (reference: >= 1000, kernel_cdist.cu:28) and no n_train*n_predict <= 67.1M cap
 * (kernel_matrixExp.cu:29); both precisions live in one library (reference: one per build,
 * CMakeLists.txt:8-12). */
int gp_predict_wrap_f64(gp_ctx* ctx, const double* expX, const double* inputs,
                        const double* invQt, const double* invQ, const double* testing,
                        double* result, double* error, double* deriv,
                        int64_t n_predict, int n_train, int n_inputs, int theta_size);
int gp_predict_wrap_f32(gp_ctx* ctx, const float* expX, const float* inputs,
                        const float* invQt, const float* invQ, const float* testing,
                        float* result, float* error, float* deriv,
                        int64_t n_predict, int n_train, int n_inputs, int theta_size);

/* predict_wrap with the gradient written ROW-MAJOR, (n_predict, n_inputs): the layout
 * GaussianProcess.gpu_predict finally returns (it transposes predict_wrap's output at
 * GaussianProcess.py:321).  Same twelve arguments; saves the caller a strided host copy. */
int gp_predict_rows_f64(gp_ctx* ctx, const double* expX, const double* inputs,
                        const double* invQt, const double* invQ, const double* testing,
                        double* result, double* error, double* deriv,
                        int64_t n_predict, int n_train, int n_inputs, int theta_size);
int gp_predict_rows_f32(gp_ctx* ctx, const float* expX, const float* inputs,
                        const float* invQt, const float* invQ, const float* testing,
                        float* result, float* error, float* deriv,
                        int64_t n_predict, int n_train, int n_inputs, int theta_size);

/* The same with float64 host arrays but FLOAT32 arithmetic on the device (what
 * gp.predict(is_gpu=True, precision=np.float32) means when the caller's arrays are float64):
 * constants are packed from the float64 values, test rows / outputs are converted while they
 * are staged, so the caller needs no float32 copies of anything. */
int gp_predict_rows_f32_h64(gp_ctx* ctx, const double* expX, const double* inputs,
                            const double* invQt, const double* invQ, const double* testing,
                            double* result, double* error, double* deriv,
                            int64_t n_predict, int n_train, int n_inputs, int theta_size);

/* ---- device-resident form --------------------------------------------------------------
 * gp_model_create_*: pack (host side, in double) and upload the per-emulator constants the
 * reference re-uploads for every block (predict.cu:17-33): sqrt(e)-scaled training inputs
 * + invQt, and invQ folded to S' in matrix-core fragment order.  compute dtype = the
 * function's dtype.  invQ may be NULL: the model then serves gp_hessian_device only and
 * gp_predict_device on it fails with GP_ERR_INVALID. */
int gp_model_create_f64(gp_ctx* ctx, const double* expX, const double* inputs,
                        const double* invQt, const double* invQ,
                        int n_train, int n_inputs, int theta_size, gp_model** out);
int gp_model_create_f32(gp_ctx* ctx, const float* expX, const float* inputs,
                        const float* invQt, const float* invQ,
                        int n_train, int n_inputs, int theta_size, gp_model** out);
/* float32 model packed from float64 constants (rounded once, after the double-precision
 * scaling and folding): what a float32 predict on float64 data should use. */
int gp_model_create_f32_h64(gp_ctx* ctx, const double* expX, const double* inputs,
                            const double* invQt, const double* invQ,
                            int n_train, int n_inputs, int theta_size, gp_model** out);
/* Batched emulators: the per-band pattern of tests/test_perband_emulator.py:22-37 (one
 * GaussianProcess per band, all on the SAME training inputs, each with its own theta,
 * invQ, invQt), which the reference can only run as a Python loop over predict_wrap.
 *   expX [E*theta_size], inputs [n_train*n_inputs] (shared), invQt [E*n_train],
 *   invQ [E*n_train*n_train].
 * gp_predict_device on such a model runs ALL emulators over the shared test rows in one
 * launch; outputs are emulator-major: mu [E][M], var [E][M], deriv [E][M*D]. */
int gp_batch_create_f64(gp_ctx* ctx, int n_emulators, const double* expX, const double* inputs,
                        const double* invQt, const double* invQ,
                        int n_train, int n_inputs, int theta_size, gp_model** out);
int gp_batch_create_f32(gp_ctx* ctx, int n_emulators, const float* expX, const float* inputs,
                        const float* invQt, const float* invQ,
                        int n_train, int n_inputs, int theta_size, gp_model** out);
int gp_batch_create_f32_h64(gp_ctx* ctx, int n_emulators, const double* expX, const double* inputs,
                            const double* invQt, const double* invQ,
                            int n_train, int n_inputs, int theta_size, gp_model** out);
int gp_model_emulators(const gp_model* model, int* n_emulators);
int gp_model_destroy(gp_model* model);
int gp_model_info(const gp_model* model, int* dtype, int* n_train, int* n_inputs,
                  int* kernel_d, int* kernel_nb);

/* One launch of the fused kernel on the context's stream (asynchronous).  d_testing,
 * d_mu, d_var, d_deriv are DEVICE pointers of the model's dtype: testing [M*D] row-major,
 * mu [M], var [M], deriv [M*D] in deriv_layout. */
int gp_predict_device(gp_ctx* ctx, const gp_model* model, const void* d_testing,
                      void* d_mu, void* d_var, void* d_deriv, int64_t n_predict,
                      int deriv_layout);

/* The same launch for HOST arrays: testing [M*D] in, result / error [E][M] and deriv [E][M*D]
 * (deriv_layout as above) out, where E = the model's number of emulators (1 unless batched).
 * This is the body of gp_predict_wrap_* / gp_predict_rows_* without the per-call constants:
 * rows flow in slabs through three slots (pinned staging, one stream each: H2D, kernel, D2H)
 * while the calling thread and the context's helper threads (GP_HOST_THREADS, default 8) copy
 * the next slab in and the previous one out of the caller's arrays.  host_dtype is the dtype of
 * the four host arrays: the model's, or GP_F64 with a GP_F32 model -- rows are then centred and
 * scaled in double and rounded once while they are staged, outputs widened on the way back.
 * max_block_rows > 0 bounds the rows per launch (GaussianProcess.gpu_predict's `threshold`,
 * gp_emulator/GaussianProcess.py:273,297-299); 0 = the library's own slab size.
 * Returns when the outputs are written. */
int gp_predict_host(gp_ctx* ctx, const gp_model* model, int host_dtype, const void* testing,
                    void* result, void* error, void* deriv, int64_t n_predict, int deriv_layout,
                    int64_t max_block_rows);
/* number of host threads (caller included) the context uses for staging copies */
int gp_ctx_host_threads(gp_ctx* ctx, int* n_threads);
/* NUMA node the device is attached to (sysfs, by PCI bus id; -1 = unknown).  The context's helper
 * threads and pinned staging live there (GP_HOST_PIN=0 disables); a caller that wants its own
 * arrays there too binds itself to that node's cpus before allocating them. */
int gp_device_numa_node(int device, int* node);

/* ---- Hessian of the mean ------------------------------------------------------------------
 * Replaces GaussianProcess.hessian (gp_emulator/GaussianProcess.py:345-366; the reference
 * has NO native version of it).  hess is (n_predict, n_inputs, n_inputs) row-major.
 * gp_hessian_device: device pointers, asynchronous on the context's stream.
 * gp_hessian_f64/_f32: host pointers in and out (invQ is not needed by the Hessian).
 * Every element and its mirror image are stored from the same value: the result is exactly
 * symmetric.  n_inputs >= 6 (with n_train <= 320) runs on the matrix core; the first Hessian
 * call on a model packs and uploads that kernel's constant operand (once, thread-safe). */
int gp_hessian_device(gp_ctx* ctx, const gp_model* model, const void* d_testing,
                      void* d_hess, int64_t n_predict);
/* host arrays of the model's dtype in and out, through the same slab pipeline as gp_predict_host */
int gp_hessian_host(gp_ctx* ctx, const gp_model* model, const void* testing, void* hess,
                    int64_t n_predict);
/* the same for float64 host arrays whatever the model's dtype: a float32 model takes the rows and
 * returns the matrices through the staging copies' conversions (what hessian(precision=float32)
 * means for a float64 caller; numpy casts of the (M, D, D) result cost more than the kernel) */
int gp_hessian_host_h64(gp_ctx* ctx, const gp_model* model, const double* testing, double* hess,
                        int64_t n_predict);
int gp_hessian_f64(gp_ctx* ctx, const double* expX, const double* inputs, const double* invQt,
                   const double* testing, double* hess,
                   int64_t n_predict, int n_train, int n_inputs, int theta_size);
int gp_hessian_f32(gp_ctx* ctx, const float* expX, const float* inputs, const float* invQt,
                   const float* testing, float* hess,
                   int64_t n_predict, int n_train, int n_inputs, int theta_size);

/* ---- multivariate reconstruction ----------------------------------------------------------
 * out[r][band] = sum_p coef[p][r] * basis[p][band]  for r < n_rows, device pointers, dtype =
 * GP_F32 / GP_F64, asynchronous on the context's stream, n_pcs <= 16.  With coef = the mean of
 * a batched predict ([n_pcs][M]) this is MultivariateEmulator.predict's reconstruction
 * (gp_emulator/multivariate_gp.py:214-216) for M rows at once; with coef = its gradient
 * ([n_pcs][M*D], rows r = (m, d)) it is the Jacobian (:218).  The reference does both on the
 * host for one test row per call. */
int gp_reconstruct_device(gp_ctx* ctx, int dtype, const void* d_basis, const void* d_coef,
                          void* d_out, int64_t n_rows, int n_pcs, int n_bands);

/* MultivariateEmulator.predict (gp_emulator/multivariate_gp.py:195-222) in one call, for the
 * caller that asks for one state vector (or a few) at a time: `model` is the batch of the n_pcs
 * per-PC emulators (gp_batch_create_*), d_basis their basis functions [n_pcs][n_bands] on the
 * device (model's dtype), y host rows [n_rows][n_inputs].  Fills host arrays fwd [n_rows][n_bands]
 * and, unless jac is NULL, jac [n_rows][n_inputs][n_bands]: rows up, batched predict,
 * reconstruction and Jacobian on the device, results down (one copy when jac == fwd +
 * n_rows * n_bands, i.e. the two arrays are laid out back to back), one synchronisation.  At most 1 GiB
 * of results per call (GP_ERR_UNSUPPORTED beyond: split the rows). */
int gp_mv_predict_host(gp_ctx* ctx, const gp_model* model, const void* d_basis, const void* y,
                       int64_t n_rows, int n_bands, void* fwd, void* jac);

/* The reference re-uploads every constant on every call (gpu/predict.cu:11-34), so whatever the caller has
 * done to theta, invQ, invQt or the basis -- in place or not -- is what the call computes with.  A device-resident
 * copy must not lose that: gp_content_digest is a 64-bit digest of the host memory blocks such a copy was made
 * from (no GPU needed; every byte counts), and gp_mv_predict_host_checked is gp_mv_predict_host that digests the
 * same blocks again WHILE THE DEVICE WORKS (the calling thread would only wait) and returns GP_STALE -- with the
 * results to be discarded and the copy to be rebuilt -- when it is no longer `expected`. */
uint64_t gp_content_digest(const void* const* blocks, const int64_t* nbytes, int n_blocks);
int gp_mv_predict_host_checked(gp_ctx* ctx, const gp_model* model, const void* d_basis, const void* y,
                               int64_t n_rows, int n_bands, void* fwd, void* jac,
                               const void* const* blocks, const int64_t* nbytes, int n_blocks, uint64_t expected);

/* ---- training objective (next after the predict path: SURVEY.md 8f rank 2) -------------------
 * For each of n_sets hyper-parameter vectors theta [n_sets][n_inputs+2]: what
 * GaussianProcess.loglikelihood + partial_devs compute (gp_emulator/GaussianProcess.py:52-125):
 * cost [n_sets], grad [n_sets][n_inputs+2], and optionally the by-products of
 * _prepare_likelihood, invQ [n_sets][n_train^2] and invQt [n_sets][n_train] (NULL to skip).
 * targets is [n_train] when targets_shared != 0 (random restarts of one emulator), else
 * [n_sets][n_train] (per-band emulators).  Host pointers, fp64 only, synchronous. */
int gp_likelihood_batch_f64(gp_ctx* ctx, int n_sets, const double* theta, const double* inputs,
                            const double* targets, int targets_shared, int n_train, int n_inputs,
                            double* cost, double* grad, double* invQ, double* invQt);

/* Host-side packing only (no GPU needed): what gp_model_create_* uploads.  xa and frags
 * are sized by gp_pack_sizes; sd takes 2*kernel_d + 1 reals (sqrt(e_d), the centre c_d, b);
 * used by the CPU tests to check the fragment layout. */
/* position (in units of 64-real fragments) of fragment (I >= J, k-step s) in the packed buffer */
int gp_frag_index(int kernel_nb, int I, int J, int s);
/* k-steps (groups of 4 training points) of the predict kernel chosen for this shape; k-steps
 * 4 I + s beyond it exist in the packed buffer (as zeros) but are never issued.  0 = the
 * general-shape kernel. */
int gp_kernel_ksteps(int n_train, int n_inputs, int* ksteps);
int gp_pack_sizes(int dtype, int n_train, int n_inputs, int* kernel_d, int* kernel_nb,
                  int64_t* xa_len, int64_t* frags_len);
int gp_pack_model_f64(const double* expX, const double* inputs, const double* invQt,
                      const double* invQ, int n_train, int n_inputs, int theta_size,
                      double* xa, double* frags, double* sd, double* b);
int gp_pack_model_f32(const float* expX, const float* inputs, const float* invQt,
                      const float* invQ, int n_train, int n_inputs, int theta_size,
                      float* xa, float* frags, float* sd, float* b);

/* Page-locked host memory for callers that keep their test rows and result arrays in it: gp_predict_host (one
 * emulator, row-major gradient, arrays of the model's precision) recognises such arrays (gp_pinned_alloc,
 * hipHostMalloc or hipHostRegister -- by hipPointerGetAttributes) and copies every slab straight between them and the
 * device: no staging, no host copies, uploads and downloads on their own streams so that both directions of the
 * link run at once.  (The reference copies out of and into the caller's pageable numpy arrays with blocking
 * cudaMemcpy, gpu/predict.cu:11-34, 73, 117, 150.)  Allocated on the NUMA node the context's device hangs off. */
int gp_pinned_alloc(gp_ctx* ctx, int64_t bytes, void** ptr);
int gp_pinned_free(gp_ctx* ctx, void* ptr);

/* ---- device memory and timing plumbing (so the Python host needs no GPU framework) --- */
int gp_malloc(gp_ctx* ctx, int64_t bytes, void** dptr);
int gp_free(gp_ctx* ctx, void* dptr);
int gp_memcpy_h2d(gp_ctx* ctx, void* dst_device, const void* src_host, int64_t bytes);
int gp_memcpy_d2h(gp_ctx* ctx, void* dst_host, const void* src_device, int64_t bytes);
int gp_memset(gp_ctx* ctx, void* dptr, int value, int64_t bytes);
int gp_event_create(gp_ctx* ctx, gp_event** out);
int gp_event_destroy(gp_event* ev);
int gp_event_record(gp_ctx* ctx, gp_event* ev);   /* on the context's stream */
int gp_event_elapsed_ms(gp_event* start, gp_event* stop, float* ms); /* syncs on stop */

#ifdef __cplusplus
}
#endif
#endif /* GP_PREDICT_HIP_H */
