"""ctypes binding of libgp_predict_hip.so (the C ABI in include/gp_predict_hip.h).

No PyTorch, no GPU framework: device memory, streams and events all go through the
library.  Loading FAILS LOUDLY (``GpuPredictUnavailable``) when the library has not been
built or no GPU is visible -- there is no CPU fallback anywhere behind ``is_gpu=True``.
"""
import ctypes
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# GP_PREDICT_LIB lets tools/ab_bench.py time differently-built variants of the library on
# one device; it never points at anything but a build of this repo's csrc/.
LIB_PATH = os.environ.get("GP_PREDICT_LIB") or os.path.join(HERE, "libgp_predict_hip.so")

GP_F32, GP_F64 = 0, 1
GP_DERIV_DMAJOR, GP_DERIV_ROWMAJOR = 0, 1

c_int, c_i64, c_void_p = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
c_dp = ctypes.POINTER(ctypes.c_double)
c_fp = ctypes.POINTER(ctypes.c_float)
PP = ctypes.POINTER(c_void_p)

# name -> (restype, argtypes); every symbol include/gp_predict_hip.h declares
SIGNATURES = {
    "gp_last_error_string": (ctypes.c_char_p, []),
    "gp_version_string": (ctypes.c_char_p, []),
    "gp_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "gp_ctx_create": (c_int, [c_int, PP]),
    "gp_ctx_destroy": (c_int, [c_void_p]),
    "gp_ctx_synchronize": (c_int, [c_void_p]),
    "gp_ctx_set_debug_buffer": (c_int, [c_void_p, c_void_p]),
    "gp_ctx_device_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_i64),
                                   ctypes.c_char_p, c_int]),
    "gp_predict_wrap_f64": (c_int, [c_void_p] + [c_void_p] * 8 + [c_i64, c_int, c_int, c_int]),
    "gp_predict_wrap_f32": (c_int, [c_void_p] + [c_void_p] * 8 + [c_i64, c_int, c_int, c_int]),
    "gp_predict_rows_f64": (c_int, [c_void_p] + [c_void_p] * 8 + [c_i64, c_int, c_int, c_int]),
    "gp_predict_rows_f32": (c_int, [c_void_p] + [c_void_p] * 8 + [c_i64, c_int, c_int, c_int]),
    "gp_predict_rows_f32_h64": (c_int, [c_void_p] + [c_void_p] * 8 + [c_i64, c_int, c_int, c_int]),
    "gp_model_create_f64": (c_int, [c_void_p] + [c_void_p] * 4 + [c_int, c_int, c_int, PP]),
    "gp_model_create_f32": (c_int, [c_void_p] + [c_void_p] * 4 + [c_int, c_int, c_int, PP]),
    "gp_model_create_f32_h64": (c_int, [c_void_p] + [c_void_p] * 4 + [c_int, c_int, c_int, PP]),
    "gp_batch_create_f32_h64": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_int, c_int, PP]),
    "gp_batch_create_f64": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_int, c_int, PP]),
    "gp_batch_create_f32": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_int, c_int, PP]),
    "gp_model_emulators": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "gp_model_destroy": (c_int, [c_void_p]),
    "gp_model_info": (c_int, [c_void_p] + [ctypes.POINTER(c_int)] * 5),
    "gp_predict_device": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_i64, c_int]),
    "gp_predict_host": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_i64, c_int, c_i64]),
    "gp_ctx_host_threads": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "gp_device_numa_node": (c_int, [c_int, ctypes.POINTER(c_int)]),
    "gp_kernel_ksteps": (c_int, [c_int, c_int, ctypes.POINTER(c_int)]),
    "gp_hessian_device": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64]),
    "gp_hessian_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64]),
    "gp_hessian_f64": (c_int, [c_void_p] + [c_void_p] * 5 + [c_i64, c_int, c_int, c_int]),
    "gp_hessian_f32": (c_int, [c_void_p] + [c_void_p] * 5 + [c_i64, c_int, c_int, c_int]),
    "gp_reconstruct_device": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int]),
    "gp_hessian_host_h64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64]),
    "gp_mv_predict_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "gp_content_digest": (ctypes.c_uint64, [c_void_p, c_void_p, c_int]),
    "gp_mv_predict_host_checked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_int, ctypes.c_uint64]),
    "gp_frag_index": (c_int, [c_int, c_int, c_int, c_int]),
    "gp_likelihood_batch_f64": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gp_pack_sizes": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                              ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "gp_pack_model_f64": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int] + [c_void_p] * 4),
    "gp_pack_model_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int] + [c_void_p] * 4),
    "gp_pinned_alloc": (c_int, [c_void_p, c_i64, PP]),
    "gp_pinned_free": (c_int, [c_void_p, c_void_p]),
    "gp_malloc": (c_int, [c_void_p, c_i64, PP]),
    "gp_free": (c_int, [c_void_p, c_void_p]),
    "gp_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_i64]),
    "gp_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_i64]),
    "gp_memset": (c_int, [c_void_p, c_void_p, c_int, c_i64]),
    "gp_event_create": (c_int, [c_void_p, PP]),
    "gp_event_destroy": (c_int, [c_void_p]),
    "gp_event_record": (c_int, [c_void_p, c_void_p]),
    "gp_event_elapsed_ms": (c_int, [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_float)]),
}


class GpuPredictUnavailable(RuntimeError):
    """The HIP library is missing or unusable.  Never caught inside this package."""


class GpuPredictError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


_lib = None
_lock = threading.Lock()


def load():
    """dlopen the library and attach signatures (no GPU is touched)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GpuPredictUnavailable(
                "%s not built; run `python -m gp_emulator_amd.build` (needs hipcc). "
                "There is no CPU fallback for is_gpu=True." % LIB_PATH)
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise GpuPredictUnavailable("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().gp_last_error_string().decode("utf-8", "replace")
        raise GpuPredictError("%s failed (status %d): %s" % (what or "gp call", rc, msg))


GP_STALE = 1     # gp_mv_predict_host_checked: the host data behind a device-resident copy has changed


class HostBlocks:
    """The host arrays a device-resident copy was made from, as the (pointer, length) table the C ABI digests
    (``gp_content_digest``: a 64-bit digest of every byte, at memory speed, no GPU needed).  ``same_arrays`` is the
    cheap pre-filter (are these still the very array objects?); ``digest()`` reads the bytes as they are NOW, so an
    in-place edit of any element shows.  Members that are not C-contiguous arrays (a strided theta column, a
    list) are digested from a contiguous buffer that ``refresh()`` re-fills from the member: their pointers in the
    table never change, and a 0.5 MB inverse is never copied unless the caller made it non-contiguous."""

    def __init__(self, arrays):
        self.arrays = list(arrays)           # the caller's objects themselves
        self.ids = tuple(id(a) for a in arrays)
        self._bufs = {}                      # index -> contiguous buffer of a non-contiguous member
        src = []
        for i, a in enumerate(self.arrays):
            if isinstance(a, np.ndarray) and a.flags.c_contiguous:
                src.append(a)
            else:
                self._bufs[i] = np.array(a, order="C", copy=True)
                src.append(self._bufs[i])
        self.n = len(src)
        self.ptrs = (c_void_p * self.n)(*[a.ctypes.data for a in src])
        self.lens = (c_i64 * self.n)(*[a.nbytes for a in src])
        self.expected = self.digest()

    def same_arrays(self, arrays):
        return self.ids == tuple(id(a) for a in arrays)

    def refresh(self):
        """Before handing ``ptrs`` to a checked call: the non-contiguous members as they are now (a member whose
        shape has changed no longer fits its buffer: the digest is then made to differ)."""
        for i, buf in self._bufs.items():
            a = np.asarray(self.arrays[i])
            if a.shape == buf.shape and a.dtype == buf.dtype:
                np.copyto(buf, a)
            else:
                buf.view(np.uint8)[...] = 0xA5
        return self

    def digest(self):
        self.refresh()
        return int(load().gp_content_digest(self.ptrs, self.lens, self.n))

    def unchanged(self):
        return self.digest() == self.expected


def device_count():
    n = c_int(0)
    rc = load().gp_device_count(ctypes.byref(n))
    if rc != 0:
        return 0
    return n.value


def _ptr(a):
    return a.ctypes.data_as(c_void_p)


def device_numa_cpus(device=0):
    """The cpus of the NUMA node ``device`` is attached to that this process may run on (empty
    set when sysfs does not tell).  Initialises the HIP runtime."""
    node = c_int(-1)
    if load().gp_device_numa_node(int(device), ctypes.byref(node)) != 0 or node.value < 0:
        return set()
    try:
        text = open("/sys/devices/system/node/node%d/cpulist" % node.value).read().strip()
    except OSError:
        return set()
    cpus = set()
    for part in text.split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus & set(os.sched_getaffinity(0))


def bind_near_device(device=0):
    """Restrict the calling process to the cpus next to ``device`` (what ``numactl
    --cpunodebind`` would do for a one-process-per-GPU launch): arrays it allocates afterwards
    are first-touched on the socket whose PCIe root the device hangs off, so neither the
    staging copies nor the DMA cross the socket interconnect.  Returns the cpu set used (empty:
    nothing changed)."""
    cpus = device_numa_cpus(device)
    if cpus:
        os.sched_setaffinity(0, cpus)
    return cpus


class OutputPool:
    """Recycles the memory of predict's output arrays.

    A fresh 100 MB numpy array costs more than the predict that fills it: every page is
    first-touch faulted while the results are copied in, and unmapped again (with TLB shootdowns
    to every helper thread's core) when the caller drops it -- 7 ms per 1e6 float64 rows on the
    GPU box against 4 ms for the whole predict.  So the arrays ``Model.predict`` returns are views
    of pooled buffers, and a buffer is handed out again once nothing but the pool refers to it
    (``sys.getrefcount``): a caller that keeps its results keeps their memory, a loop that drops
    them runs on warm pages.  Buffers over ``max_item`` bytes are never pooled and the pool holds
    at most ``max_total`` bytes."""

    def __init__(self, max_item=256 << 20, max_total=1 << 30):
        self.max_item, self.max_total = max_item, max_total
        self.items = []

    def take(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < (1 << 20) or nbytes > self.max_item:
            return np.empty(shape, dtype)
        base = None
        for cand in self.items:
            # references to a free buffer: the pool's list, ``cand`` and getrefcount's argument
            if cand.nbytes == nbytes and sys.getrefcount(cand) == 3:
                base = cand
                break
        cand = None
        if base is None:
            base = np.empty(nbytes, np.uint8)
            self.items.append(base)
            total = sum(b.nbytes for b in self.items)
            # over budget: forget buffers (free ones first); memory still in use stays alive through
            # its users' references and is simply no longer recycled
            for free_first in (True, False):
                i = 0
                while total > self.max_total and i < len(self.items):
                    b = self.items[i]
                    if b is not base and (not free_first or sys.getrefcount(b) == 3):
                        total -= b.nbytes
                        del self.items[i]
                    else:
                        i += 1
                b = None
        return base.view(dtype).reshape(shape)


class Context:
    """One device + one HIP stream (gp_ctx).  Use one per thread / per GPU."""

    def __init__(self, device=0):
        self.lib = load()
        h = c_void_p()
        rc = self.lib.gp_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            msg = self.lib.gp_last_error_string().decode("utf-8", "replace")
            raise GpuPredictUnavailable("no usable GPU context on device %d: %s" % (device, msg))
        self.h = h
        self.device = int(device)
        self.out_pool = OutputPool()

    def pinned_empty(self, shape, dtype=np.float64):
        """An uninitialised numpy array in page-locked host memory (``gp_pinned_alloc``; freed with the array).
        ``Model.predict`` / ``GaussianProcess.predict(is_gpu=True)`` on test rows AND ``out=`` arrays that all live
        in such memory (model's precision, one emulator, row-major gradient) copies every slab straight between
        the arrays and the device -- no staging, no host copies, both directions of the link at once."""
        import weakref
        dtype = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64))
        nbytes = max(1, n * dtype.itemsize)
        ptr = c_void_p()
        check(self.lib.gp_pinned_alloc(self.h, nbytes, ctypes.byref(ptr)), "gp_pinned_alloc")
        buf = (ctypes.c_char * nbytes).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=np.uint8, count=n * dtype.itemsize).view(dtype).reshape(shape)
        lib, h, addr = self.lib, self.h, ptr.value
        weakref.finalize(buf, lambda: lib.gp_pinned_free(h, c_void_p(addr)))   # when the last view of it is gone
        return arr

    def close(self):
        if getattr(self, "h", None):
            self.lib.gp_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def synchronize(self):
        check(self.lib.gp_ctx_synchronize(self.h), "gp_ctx_synchronize")

    def device_info(self):
        cu, mem = c_int(0), c_i64(0)
        name = ctypes.create_string_buffer(256)
        check(self.lib.gp_ctx_device_info(self.h, ctypes.byref(cu), ctypes.byref(mem), name, 256))
        return dict(compute_units=cu.value, hbm_bytes=mem.value, name=name.value.decode())

    def host_threads(self):
        n = c_int(0)
        check(self.lib.gp_ctx_host_threads(self.h, ctypes.byref(n)), "gp_ctx_host_threads")
        return n.value

    # ---- memory -----------------------------------------------------------------
    def malloc(self, nbytes):
        p = c_void_p()
        check(self.lib.gp_malloc(self.h, int(nbytes), ctypes.byref(p)), "gp_malloc")
        return p

    def free(self, p):
        check(self.lib.gp_free(self.h, p), "gp_free")

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(max(arr.nbytes, 1))
        if arr.nbytes:
            check(self.lib.gp_memcpy_h2d(self.h, p, _ptr(arr), arr.nbytes), "gp_memcpy_h2d")
        return p

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        check(self.lib.gp_memcpy_h2d(self.h, dptr, _ptr(arr), arr.nbytes), "gp_memcpy_h2d")

    def to_host(self, dptr, shape, dtype):
        out = self.out_pool.take(shape, dtype)      # >= 1 MB: recycled memory (see OutputPool)
        if out.nbytes:
            check(self.lib.gp_memcpy_d2h(self.h, _ptr(out), dptr, out.nbytes), "gp_memcpy_d2h")
        return out

    def to_host_at(self, dptr, offset_bytes, shape, dtype):
        """Copy ``shape`` elements starting ``offset_bytes`` into a device buffer (spot checks of
        results too large to bring back whole)."""
        out = np.empty(shape, dtype=dtype)
        src = c_void_p(dptr.value + int(offset_bytes))
        check(self.lib.gp_memcpy_d2h(self.h, _ptr(out), src, out.nbytes), "gp_memcpy_d2h")
        return out

    def reconstruct_device(self, dtype, d_basis, d_coef, d_out, n_rows, n_pcs, n_bands):
        """out[r][band] = sum_p coef[p][r] * basis[p][band] on the device (asynchronous)."""
        code = GP_F64 if np.dtype(dtype) == np.float64 else GP_F32
        check(self.lib.gp_reconstruct_device(self.h, code, d_basis, d_coef, d_out, int(n_rows),
                                             int(n_pcs), int(n_bands)), "gp_reconstruct_device")

    def likelihood_batch(self, thetas, inputs, targets, want_inverse=False):
        """cost (E,), grad (E, D+2) [and invQ (E, N, N), invQt (E, N)] of the training
        objective for E hyper-parameter sets; targets (N,) shared or (E, N)."""
        thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        targets = np.ascontiguousarray(targets, dtype=np.float64)
        E, N, D = thetas.shape[0], inputs.shape[0], inputs.shape[1]
        if thetas.shape[1] != D + 2:
            raise ValueError("theta needs n_inputs + 2 entries")
        shared = targets.ndim == 1
        if targets.shape[-1] != N or (not shared and targets.shape[0] != E):
            raise ValueError("targets must be (n_train,) or (n_sets, n_train)")
        cost = np.empty(E)
        grad = np.empty((E, D + 2))
        invQ = np.empty((E, N, N)) if want_inverse else None
        invQt = np.empty((E, N)) if want_inverse else None
        check(self.lib.gp_likelihood_batch_f64(
            self.h, E, _ptr(thetas), _ptr(inputs), _ptr(targets), int(shared), N, D, _ptr(cost),
            _ptr(grad), _ptr(invQ) if want_inverse else None,
            _ptr(invQt) if want_inverse else None), "gp_likelihood_batch_f64")
        return (cost, grad, invQ, invQt) if want_inverse else (cost, grad)

    # ---- events -----------------------------------------------------------------
    def event(self):
        e = c_void_p()
        check(self.lib.gp_event_create(self.h, ctypes.byref(e)), "gp_event_create")
        return e

    def record(self, ev):
        check(self.lib.gp_event_record(self.h, ev), "gp_event_record")

    def elapsed_ms(self, e0, e1):
        ms = ctypes.c_float(0)
        check(self.lib.gp_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "gp_event_elapsed_ms")
        return ms.value

    def event_destroy(self, ev):
        self.lib.gp_event_destroy(ev)


class Model:
    """Per-emulator constants packed and resident in HBM (gp_model)."""

    def __init__(self, ctx, expX, inputs, invQt, invQ, precision=np.float64):
        self.ctx = ctx
        self.dtype = np.dtype(precision)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("precision must be float32 or float64, got %r" % (precision,))
        # constants cross the boundary as float64 whatever the compute type: a float32 model is
        # packed from the float64 values and rounded once (gp_model_create_f32_h64)
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        if inputs.ndim != 2:
            raise ValueError("inputs must be (n_train, n_inputs)")
        self.n_train, self.n_inputs = inputs.shape
        expX = np.ascontiguousarray(expX, dtype=np.float64).ravel()
        invQt = np.ascontiguousarray(invQt, dtype=np.float64).ravel()
        if invQt.size != self.n_train:
            raise ValueError("invQt size does not match n_train")
        if invQ is not None:             # None: Hessian-only model (no variance operand)
            invQ = np.ascontiguousarray(invQ, dtype=np.float64)
            if invQ.size != self.n_train ** 2:
                raise ValueError("invQ size does not match n_train")
        fn = ctx.lib.gp_model_create_f64 if self.dtype == np.float64 else ctx.lib.gp_model_create_f32_h64
        h = c_void_p()
        check(fn(ctx.h, _ptr(expX), _ptr(inputs), _ptr(invQt),
                 _ptr(invQ) if invQ is not None else None,
                 self.n_train, self.n_inputs, expX.size, ctypes.byref(h)), "gp_model_create")
        self.h = h

    def info(self):
        v = [c_int(0) for _ in range(5)]
        check(self.ctx.lib.gp_model_info(self.h, *[ctypes.byref(x) for x in v]))
        nk = c_int(0)
        check(self.ctx.lib.gp_kernel_ksteps(v[1].value, v[2].value, ctypes.byref(nk)))
        return dict(dtype=v[0].value, n_train=v[1].value, n_inputs=v[2].value,
                    kernel_d=v[3].value, kernel_nb=v[4].value, kernel_nk=nk.value)

    def predict_device(self, d_testing, d_mu, d_var, d_deriv, n_predict,
                       deriv_layout=GP_DERIV_ROWMAJOR):
        """Asynchronous launch on the context's stream; all pointers are device pointers."""
        check(self.ctx.lib.gp_predict_device(self.ctx.h, self.h, d_testing, d_mu, d_var, d_deriv,
                                             int(n_predict), int(deriv_layout)), "gp_predict_device")

    def predict(self, testing, deriv_layout=GP_DERIV_ROWMAJOR, out=None, max_block_rows=0):
        """Host arrays in, host arrays out through the library's slab pipeline
        (``gp_predict_host``: pinned staging, three slots, helper threads).

        ``testing`` is (M, D) of the model's dtype, or float64 for a float32 model (rows are
        then centred and scaled in double while they are staged, and the outputs come back as
        float64).  ``out=(mu, var, deriv)`` supplies the output arrays (contiguous, dtype of
        ``testing``); without it the arrays come from the context's ``OutputPool`` (memory of
        results the caller has dropped is reused).  ``max_block_rows`` > 0 bounds the rows per
        launch."""
        testing = np.asarray(testing)
        if testing.dtype != np.float64 or self.dtype == np.float64:
            testing = np.ascontiguousarray(testing, dtype=self.dtype)
        else:
            testing = np.ascontiguousarray(testing)
        hdt = testing.dtype
        if testing.ndim != 2:
            raise ValueError("testing must be (n_predict, n_inputs)")
        M, D = testing.shape
        if D != self.n_inputs:
            raise ValueError("testing has %d columns, model has %d inputs" % (D, self.n_inputs))
        E = getattr(self, "n_emulators", None)
        lead = () if E is None else (E,)
        dshape = lead + ((M, D) if deriv_layout == GP_DERIV_ROWMAJOR else (D, M))
        if out is None:
            # one buffer, three views: the library brings a small call's results back in one copy
            # when result | error | deriv lie back to back
            n_e = (E or 1) * M
            flat = self.ctx.out_pool.take((n_e * (2 + D),), hdt)
            mu, var = flat[:n_e].reshape(lead + (M,)), flat[n_e:2 * n_e].reshape(lead + (M,))
            deriv = flat[2 * n_e:].reshape(dshape)
        else:
            mu, var, deriv = out
            for a, shape in ((mu, lead + (M,)), (var, lead + (M,)), (deriv, dshape)):
                if (not isinstance(a, np.ndarray) or a.dtype != hdt or a.shape != shape
                        or not a.flags["C_CONTIGUOUS"] or not a.flags["WRITEABLE"]):
                    raise ValueError("out arrays must be writeable C-contiguous %s arrays of shapes "
                                     "%s, %s, %s" % (hdt, lead + (M,), lead + (M,), dshape))
        if M:
            check(self.ctx.lib.gp_predict_host(
                self.ctx.h, self.h, GP_F64 if hdt == np.float64 else GP_F32, _ptr(testing), _ptr(mu),
                _ptr(var), _ptr(deriv), M, int(deriv_layout), int(max_block_rows)), "gp_predict_host")
        return mu, var, deriv

    def hessian_device(self, d_testing, d_hess, n_predict):
        """Asynchronous Hessian launch; device pointers, hess is (n_predict, D, D)."""
        check(self.ctx.lib.gp_hessian_device(self.ctx.h, self.h, d_testing, d_hess,
                                             int(n_predict)), "gp_hessian_device")

    def hessian(self, testing, out=None):
        """(M, D, D) Hessian of the mean for host rows, through the slab pipeline.  Rows of the
        model's dtype give matrices of that dtype; float64 rows on a float32 model are converted
        while they are staged and the matrices come back as float64 (``gp_hessian_host_h64``)."""
        testing = np.asarray(testing)
        h64 = testing.dtype == np.float64 and self.dtype != np.float64
        hdt = np.dtype(np.float64) if h64 else self.dtype
        testing = np.ascontiguousarray(testing, dtype=hdt)
        M, D = testing.shape
        if D != self.n_inputs:
            raise ValueError("testing has %d columns, model has %d inputs" % (D, self.n_inputs))
        hess = self.ctx.out_pool.take((M, D, D), hdt) if out is None else out
        if hess.shape != (M, D, D) or hess.dtype != hdt or not hess.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous (M, D, D) %s array" % hdt)
        if M:
            fn = self.ctx.lib.gp_hessian_host_h64 if h64 else self.ctx.lib.gp_hessian_host
            check(fn(self.ctx.h, self.h, _ptr(testing), _ptr(hess), M), "gp_hessian_host")
        return hess

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.gp_model_destroy(self.h)
            self.h = None

    __del__ = close


class BatchModel(Model):
    """E emulators on the SAME training inputs (per-band pattern,
    tests/test_perband_emulator.py:22-37), predicted over shared test rows in ONE launch.

    expX (E, D+2), inputs (N, D), invQt (E, N), invQ (E, N, N).
    ``predict`` (inherited: the slab pipeline) returns mu (E, M), var (E, M), deriv (E, M, D).
    """

    def __init__(self, ctx, expX, inputs, invQt, invQ, precision=np.float64):
        self.ctx = ctx
        self.dtype = np.dtype(precision)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("precision must be float32 or float64, got %r" % (precision,))
        inputs = np.ascontiguousarray(inputs, dtype=np.float64)
        self.n_train, self.n_inputs = inputs.shape
        expX = np.ascontiguousarray(expX, dtype=np.float64)
        if expX.ndim != 2:
            raise ValueError("expX must be (n_emulators, theta_size)")
        self.n_emulators = E = expX.shape[0]
        invQt = np.ascontiguousarray(invQt, dtype=np.float64)
        invQ = np.ascontiguousarray(invQ, dtype=np.float64)
        if invQt.shape != (E, self.n_train) or invQ.shape != (E, self.n_train, self.n_train):
            raise ValueError("invQt must be (E, N) and invQ (E, N, N)")
        fn = ctx.lib.gp_batch_create_f64 if self.dtype == np.float64 else ctx.lib.gp_batch_create_f32_h64
        h = c_void_p()
        check(fn(ctx.h, E, _ptr(expX), _ptr(inputs), _ptr(invQt), _ptr(invQ),
                 self.n_train, self.n_inputs, expX.shape[1], ctypes.byref(h)), "gp_batch_create")
        self.h = h

    def hessian(self, testing, out=None):
        raise GpuPredictError("hessian is per emulator; build a Model for the emulator wanted")


def pack_model(expX, inputs, invQt, invQ, precision=np.float64):
    """Host-only packing (no GPU): returns dict(xa, frags, sd, b, kernel_d, kernel_nb)."""
    lib = load()
    dt = np.dtype(precision)
    inputs = np.ascontiguousarray(inputs, dtype=dt)
    N, D = inputs.shape
    expX = np.ascontiguousarray(expX, dtype=dt).ravel()
    invQt = np.ascontiguousarray(invQt, dtype=dt).ravel()
    invQ = np.ascontiguousarray(invQ, dtype=dt)
    kd, knb, xl, fl = c_int(0), c_int(0), c_i64(0), c_i64(0)
    check(lib.gp_pack_sizes(GP_F64 if dt == np.float64 else GP_F32, N, D, ctypes.byref(kd),
                            ctypes.byref(knb), ctypes.byref(xl), ctypes.byref(fl)), "gp_pack_sizes")
    xa = np.zeros(xl.value, dt)
    fr = np.zeros(fl.value, dt)
    sd = np.zeros(2 * kd.value + 1, dt)
    b = np.zeros(1, dt)
    fn = lib.gp_pack_model_f64 if dt == np.float64 else lib.gp_pack_model_f32
    check(fn(_ptr(expX), _ptr(inputs), _ptr(invQt), _ptr(invQ), N, D, expX.size,
             _ptr(xa), _ptr(fr), _ptr(sd), _ptr(b)), "gp_pack_model")
    nk = c_int(0)
    check(lib.gp_kernel_ksteps(N, D, ctypes.byref(nk)), "gp_kernel_ksteps")
    return dict(xa=xa, frags=fr, sd=sd[:kd.value], centre=sd[kd.value:2 * kd.value], b=b[0],
                kernel_d=kd.value, kernel_nb=knb.value, kernel_nk=nk.value)


_tls = threading.local()


_default_device = [None]


def set_default_device(device):
    """The device the drop-in entry points (``predict(is_gpu=True)``, ``predict_wrap``, training with
    ``is_gpu=True``) use in this process.  The reference always computes on device 0
    (``gpu_predict.h``: no device selection at all); a one-process-per-GPU launch calls this once
    with its local rank.  Initial value: ``$GP_DEVICE`` or 0."""
    _default_device[0] = int(device)


def default_device():
    if _default_device[0] is None:
        _default_device[0] = int(os.environ.get("GP_DEVICE", "0"))
    return _default_device[0]


def default_context(device=None):
    """Per-thread context per device for the drop-in entry points (a gp_ctx owns a stream
    and a scratch buffer, so it is not shared between threads)."""
    device = default_device() if device is None else int(device)
    d = getattr(_tls, "ctx", None)
    if d is None:
        d = _tls.ctx = {}
    c = d.get(device)
    if c is None:
        c = d[device] = Context(device)
    return c
