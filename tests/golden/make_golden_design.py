#!/usr/bin/env python3
"""Generate tests/golden/design.npz from the REFERENCE's own lhd and k-fold helpers.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_design.py

As in make_golden.py the reference's Python-2 modules are translated in memory with lib2to3 and
exec'd; only numeric outputs are stored.  Cases (numpy.random seeded before each call):
  lhd_uniform   seed 7   lhd(dist=uniform(-1, 2), size=5)                (gp_emulator/lhd.py:59-66)
  lhd_norm      seed 8   lhd(dist=norm(0, 1), size=7, dims=3)            (:71-79)
  lhd_multi     seed 9   lhd(dist=(norm, beta(2,5), expon(scale=1/1.5)), size=6)   (:85-94)
  kfold_*       k_fold_cross_validation(range(10), 3)                    (GaussianProcess.py:9-26)
"""
import os
import sys
import types
import warnings

import numpy as np
import scipy.stats as ss

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GP_REFERENCE", "/root/reference")


def load(name):
    warnings.simplefilter("ignore")
    from lib2to3 import refactor
    fixers = [f for f in refactor.get_fixers_from_package("lib2to3.fixes")
              if not f.endswith("fix_import")]
    tool = refactor.RefactoringTool(fixers)
    sys.modules.setdefault("_gpu_predict", types.ModuleType("_gpu_predict"))
    path = os.path.join(REF, "gp_emulator", name + ".py")
    with open(path) as fh:
        src = fh.read().expandtabs(8)
    if not src.endswith("\n"):
        src += "\n"
    mod = types.ModuleType(name)
    mod.__file__ = path
    exec(compile(str(tool.refactor_string(src, path)), path, "exec"), mod.__dict__)
    return mod


def main():
    lhd = load("lhd").lhd
    kfold = load("GaussianProcess").k_fold_cross_validation
    out = {}
    np.random.seed(7)
    out["lhd_uniform"] = lhd(dist=ss.uniform(loc=-1, scale=2), size=5)
    np.random.seed(8)
    out["lhd_norm"] = lhd(dist=ss.norm(loc=0, scale=1), size=7, dims=3)
    np.random.seed(9)
    out["lhd_multi"] = lhd(dist=(ss.norm(loc=0, scale=1), ss.beta(2, 5), ss.expon(scale=1 / 1.5)), size=6)
    folds = list(kfold(range(10), 3))
    for k, (tr, va) in enumerate(folds):
        out["kfold_train_%d" % k] = np.array(tr)
        out["kfold_valid_%d" % k] = np.array(va)
    np.savez_compressed(os.path.join(HERE, "design.npz"), **out)
    for k, v in out.items():
        print(k, v.shape)


if __name__ == "__main__":
    main()
