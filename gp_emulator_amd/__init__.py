"""gp_emulator_amd: the GP predict hot path of UCL/gp_emulator on MI355X (gfx950).

Public names follow the reference package (gp_emulator/__init__.py:1-4) for the part of
it this package covers: ``GaussianProcess`` and ``MultivariateEmulator`` (predict side).  The HIP library is loaded
lazily, on the first ``is_gpu=True`` call; importing this package never touches the GPU.
"""
from .GaussianProcess import GaussianProcess  # noqa: F401
from .multivariate_gp import MultivariateEmulator  # noqa: F401

__all__ = ["GaussianProcess", "MultivariateEmulator"]
