"""Per-band emulators predicted together.

The reference's per-band pattern (tests/test_perband_emulator.py:22-37) builds one
``GaussianProcess`` per spectral band on the SAME training inputs ``y_train`` -- each with
its own targets, theta, invQ, invQt -- and then calls ``gp.predict(X, is_gpu=True)`` band by
band (:48-50).  ``predict_bands`` runs all of them over the shared test rows in one launch
of the fused kernel (BASELINE config 3: 2101 bands x N_train=250 x D=11, shared test set).
"""
import numpy as np

from . import _lib


def make_batch(gps, precision=np.float64, device=0):
    """Pack a list of GaussianProcess objects sharing ``inputs`` into one BatchModel."""
    if not gps:
        raise ValueError("need at least one GaussianProcess")
    inputs = np.asarray(gps[0].inputs)
    for gp in gps[1:]:
        other = np.asarray(gp.inputs)
        if other.shape != inputs.shape or not np.array_equal(other, inputs):
            raise ValueError("per-band emulators must share the same training inputs")
    expX = np.stack([np.exp(gp.theta) for gp in gps])
    invQt = np.stack([np.asarray(gp.invQt) for gp in gps])
    invQ = np.stack([np.asarray(gp.invQ) for gp in gps])
    return _lib.BatchModel(_lib.default_context(device), expX, inputs, invQt, invQ, precision)


def predict_bands(gps, testing, precision=np.float64, device=0):
    """mu (E, M), var (E, M), deriv (E, M, D) for the E emulators in ``gps``."""
    batch = make_batch(gps, precision, device)
    try:
        return batch.predict(np.asarray(testing))
    finally:
        batch.close()
