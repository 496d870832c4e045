"""gp_emulator_amd: the GP predict hot path of UCL/gp_emulator on MI355X (gfx950).

Public names follow the reference package (gp_emulator/__init__.py:1-4): ``GaussianProcess``,
``k_fold_cross_validation``, ``MultivariateEmulator``, ``lhd``, ``EmulatorStorage``.  The HIP
library is loaded lazily, on the first ``is_gpu=True`` call; importing this package never
touches the GPU.
"""
from .GaussianProcess import GaussianProcess, k_fold_cross_validation  # noqa: F401
from .multivariate_gp import MultivariateEmulator  # noqa: F401
from .lhd import lhd  # noqa: F401
from .save_emulators import EmulatorStorage  # noqa: F401
from ._lib import set_default_device  # noqa: F401  (ctypes only: loads nothing until first use)


def pinned_empty(shape, dtype="float64", device=None):
    """An uninitialised numpy array in page-locked host memory next to ``device`` (default: the default device).
    Test rows and ``out=`` arrays kept in such memory go to and from the device without staging copies."""
    from . import _lib
    return _lib.default_context(device).pinned_empty(shape, dtype)


__all__ = ["GaussianProcess", "k_fold_cross_validation", "MultivariateEmulator", "lhd",
           "EmulatorStorage", "set_default_device", "pinned_empty"]
