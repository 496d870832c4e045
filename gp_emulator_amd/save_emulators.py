"""A file of named emulators.

Counterpart of gp_emulator/save_emulators.py:18-106 (``EmulatorStorage``): the same three
methods -- ``dump_emulator(emulator, tag)``, ``get_keys()``, ``get_emulator(tag)`` -- over
either kind of emulator, but

* the container is an ``.npz`` archive of plain arrays (read back with
  ``allow_pickle=False``), not a ``shelve`` pickle: nothing in the file is executable;
* the two defects that keep the reference's class from round-tripping are not reproduced:
  it stores a scalar emulator's inputs under ``"input"`` and reads ``"inputs"`` (:56 vs :101),
  and hands ``MultivariateEmulator`` a ``basis_functions`` keyword it does not take (:93-98);
  tags are normalised the same way when writing and when reading (the reference writes
  ``repr(tag)`` but looks up ``repr(tuple(tag))``, :45 vs :74);
* ``get_emulator(tag, is_gpu=True)`` rebuilds invQ / invQt with the HIP likelihood kernel
  (one launch for all the principal components of a multivariate emulator) instead of host
  LAPACK, so a stored emulator goes from disk to a device-resident model without a host
  inverse.

What is stored is what the reference stores: training inputs, targets and theta for a
``GaussianProcess``; X, y, basis functions, n_pcs, thresh and hyperparams for a
``MultivariateEmulator`` (:48-60).
"""
import os

import numpy as np

from .GaussianProcess import GaussianProcess
from .multivariate_gp import MultivariateEmulator

__all__ = ["EmulatorStorage"]

_SCALAR_FIELDS = ("inputs", "targets", "theta")
_MULTI_FIELDS = ("X", "y", "basis_functions", "n_pcs", "thresh", "hyperparams")


def _tag_string(tag):
    if isinstance(tag, str):
        return tag
    try:
        return repr(tuple(tag))
    except TypeError:
        return repr(tag)


class EmulatorStorage(object):
    def __init__(self, fname):
        self.fname = fname

    # ---- archive <-> dict of arrays ------------------------------------------------------
    def _read(self):
        if not os.path.exists(self.fname):
            raise IOError("File %s doesn't exist!" % self.fname)
        with np.load(self.fname, allow_pickle=False) as f:
            return {k: f[k] for k in f.files}

    def _write(self, arrays):
        tmp = self.fname + ".tmp"
        with open(tmp, "wb") as fh:                    # a handle: numpy must not append ".npz"
            np.savez_compressed(fh, **arrays)
        os.replace(tmp, self.fname)

    @staticmethod
    def _tags(arrays):
        return [str(t) for t in arrays.get("__tags__", np.array([], dtype=str))]

    # ---- the reference's interface -------------------------------------------------------
    def dump_emulator(self, emulator, tag):
        """Add (or replace) ``emulator`` under ``tag`` (reference :22-63)."""
        arrays = self._read() if os.path.exists(self.fname) else {}
        tags = self._tags(arrays)
        tag = _tag_string(tag)
        if isinstance(emulator, MultivariateEmulator):
            kind = "multivariate"
            fields = {"X": emulator.X_train, "y": emulator.y_train,
                      "basis_functions": emulator.basis_functions, "n_pcs": emulator.n_pcs,
                      "thresh": emulator.thresh, "hyperparams": emulator.hyperparams}
        elif isinstance(emulator, GaussianProcess):
            kind = "scalar"
            fields = {"inputs": emulator.inputs, "targets": emulator.targets,
                      "theta": emulator.theta}
        else:
            raise TypeError("expected a GaussianProcess or a MultivariateEmulator")
        if tag in tags:
            slot = tags.index(tag)
            for k in [k for k in arrays if k.startswith("e%d_" % slot)]:
                del arrays[k]
        else:
            slot = len(tags)
            tags.append(tag)
        arrays["__tags__"] = np.array(tags, dtype=str)
        arrays["e%d_kind" % slot] = np.array(kind)
        for name, value in fields.items():
            arrays["e%d_%s" % (slot, name)] = np.asarray(value)
        self._write(arrays)

    def get_keys(self):
        """The stored tags (reference :65-75)."""
        return self._tags(self._read())

    def get_emulator(self, tag, is_gpu=False):
        """Rebuild the emulator stored under ``tag`` (reference :80-106)."""
        arrays = self._read()
        tags = self._tags(arrays)
        tag = _tag_string(tag)
        if tag not in tags:
            raise KeyError(tag)
        slot = tags.index(tag)

        def field(name):
            return arrays["e%d_%s" % (slot, name)]
        if str(field("kind")) == "multivariate":
            return MultivariateEmulator(X=field("X"), y=field("y"),
                                        hyperparams=field("hyperparams"),
                                        thresh=float(field("thresh")),
                                        basis_functions=field("basis_functions"),
                                        n_pcs=int(field("n_pcs")), is_gpu=is_gpu)
        gp = GaussianProcess(field("inputs"), field("targets"))
        gp._set_params(field("theta"), is_gpu=is_gpu)
        return gp
