"""Drop-in for the reference's ``_gpu_predict`` extension module.

The reference module is a CPython-2 extension with ONE function, ``predict_wrap``
(gp_emulator/gpu/_gpu_predict.cpp:15-31,115-159), taking twelve positionals parsed as
``"O!O!O!O!O!O!O!O!iiii"`` (:86-95) and returning ``None`` after filling ``result``,
``error`` and ``deriv`` in place.  This shim keeps the name, the argument order and
meaning, the in-place outputs and the dimension-major ``deriv`` layout
(predict.cu:139-150), and forwards to ``gp_predict_wrap_f32/_f64`` of
libgp_predict_hip.so through ctypes.

Behavioural differences, all deliberate:
  * dtype is dispatched at run time (float32 or float64 arrays); the reference is built
    for one precision and calls ``exit()`` on the other (_gpu_predict.cpp:39-58).
    Mixed / unsupported dtypes, non-1-D or non-contiguous arrays raise ``TypeError`` /
    ``ValueError`` instead of killing the process;
  * no minimum ``n_predict`` (reference: 1000, kernel_cdist.cu:28) and no
    ``n_train*n_predict`` cap (kernel_matrixExp.cu:29);
  * no CPU fallback: a missing library or GPU raises ``GpuPredictUnavailable``.
"""
import numpy as np

from . import _lib

_NAMES = ("expX", "inputs", "invQt", "invQ", "testing", "result", "error", "deriv")


def predict_wrap(expX, inputs, invQt, invQ, testing, result, error, deriv,
                 n_predict, n_train, n_inputs, theta_size):
    """The reference entry point: ``deriv`` comes back dimension-major."""
    return _call("wrap", expX, inputs, invQt, invQ, testing, result, error, deriv,
                 n_predict, n_train, n_inputs, theta_size)


def predict_rows(expX, inputs, invQt, invQ, testing, result, error, deriv,
                 n_predict, n_train, n_inputs, theta_size):
    """Same twelve arguments, but ``deriv`` is written row-major ``(n_predict, n_inputs)``
    (flattened) -- what ``gpu_predict`` returns after its transpose (reference :321) -- so
    the host needs no strided copy.  Not part of the reference module."""
    return _call("rows", expX, inputs, invQt, invQ, testing, result, error, deriv,
                 n_predict, n_train, n_inputs, theta_size)


def predict_rows_f32_h64(expX, inputs, invQt, invQ, testing, result, error, deriv,
                         n_predict, n_train, n_inputs, theta_size):
    """``predict_rows`` for float64 arrays with float32 arithmetic on the device: the library
    converts while it stages, so ``gpu_predict(precision=np.float32)`` needs no float32 copies
    of its (float64) inputs and outputs.  Not part of the reference module."""
    return _call("rows_f32_h64", expX, inputs, invQt, invQ, testing, result, error, deriv,
                 n_predict, n_train, n_inputs, theta_size)


def _call(kind, expX, inputs, invQt, invQ, testing, result, error, deriv,
          n_predict, n_train, n_inputs, theta_size):
    arrays = (expX, inputs, invQt, invQ, testing, result, error, deriv)
    for name, a in zip(_NAMES, arrays):
        if not isinstance(a, np.ndarray):
            raise TypeError("%s must be a numpy array" % name)   # "O!" with PyArray_Type
        if a.ndim != 1:
            raise ValueError("%s must be a 1-D vector (checkRealVector)" % name)
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("%s must be contiguous" % name)
    dt = expX.dtype
    if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise TypeError("arrays must be float32 or float64, got %s" % dt)
    for name, a in zip(_NAMES, arrays):
        if a.dtype != dt:
            raise TypeError("%s is %s but expX is %s: all arrays must share one precision"
                            % (name, a.dtype, dt))
    n_predict, n_train = int(n_predict), int(n_train)
    n_inputs, theta_size = int(n_inputs), int(theta_size)
    need = dict(expX=theta_size, inputs=n_train * n_inputs, invQt=n_train,
                invQ=n_train * n_train, testing=n_predict * n_inputs, result=n_predict,
                error=n_predict, deriv=n_predict * n_inputs)
    for name, a in zip(_NAMES, arrays):
        if a.size < need[name]:
            raise ValueError("%s has %d elements, needs %d" % (name, a.size, need[name]))
    for name in ("result", "error", "deriv"):
        if not dict(zip(_NAMES, arrays))[name].flags["WRITEABLE"]:
            raise ValueError("%s must be writeable" % name)
    ctx = _lib.default_context()
    if kind == "rows_f32_h64":
        if dt != np.float64:
            raise TypeError("predict_rows_f32_h64 takes float64 arrays")
        name = "gp_predict_rows_f32_h64"
    else:
        name = "gp_predict_%s_%s" % (kind, "f64" if dt == np.float64 else "f32")
    fn = getattr(ctx.lib, name)
    p = [a.ctypes.data_as(_lib.c_void_p) for a in arrays]
    _lib.check(fn(ctx.h, *p, n_predict, n_train, n_inputs, theta_size), "gp_predict_wrap")
    return None
