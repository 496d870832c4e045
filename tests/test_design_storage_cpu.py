"""SURVEY.md 8(f) ranks 3 and 4 on the host: Latin-hypercube designs and k-fold partitions
against the reference's own outputs (tests/golden/design.npz, made by
tests/golden/make_golden_design.py from gp_emulator/lhd.py and GaussianProcess.py:9-26), and the
EmulatorStorage round trip (gp_emulator/save_emulators.py) on the numpy branch."""
import os

import numpy as np
import pytest
import scipy.stats as ss

from conftest import load_golden
from gp_emulator_amd import (EmulatorStorage, GaussianProcess, MultivariateEmulator,
                             k_fold_cross_validation, lhd)


def test_lhd_matches_reference_draw_for_draw():
    g = load_golden("design")
    np.random.seed(7)
    assert np.array_equal(lhd(dist=ss.uniform(loc=-1, scale=2), size=5), g["lhd_uniform"])
    np.random.seed(8)
    assert np.array_equal(lhd(dist=ss.norm(loc=0, scale=1), size=7, dims=3), g["lhd_norm"])
    np.random.seed(9)
    d = (ss.norm(loc=0, scale=1), ss.beta(2, 5), ss.expon(scale=1 / 1.5))
    assert np.array_equal(lhd(dist=d, size=6), g["lhd_multi"])


@pytest.mark.parametrize("form", ["randomized", "spacefilling"])
def test_lhd_is_a_latin_hypercube(form):
    """One sample in each of `size` equal-probability strata, in every column."""
    np.random.seed(3)
    size, dims = 40, 4
    x = lhd(dist=ss.uniform(loc=0, scale=1), size=size, dims=dims, form=form, iterations=5)
    assert x.shape == (size, dims)
    for c in range(dims):
        assert sorted(np.floor(x[:, c] * size).astype(int)) == list(range(size))


def test_lhd_edge_cases():
    assert lhd(dist=None, size=5) is None
    assert lhd(dist=ss.norm(), size=None) is None
    with pytest.raises(NotImplementedError):
        lhd(dist=ss.norm(), size=4, form="orthogonal")
    with pytest.raises(ValueError):
        lhd(dist=ss.norm(), size=4, form="nope")
    with pytest.raises(AssertionError):
        lhd(dist=ss.norm(), size=4, dims=0)
    np.random.seed(0)
    a = lhd(dist=ss.uniform(), size=30, dims=3, form="spacefilling", iterations=20)
    np.random.seed(0)
    b = lhd(dist=ss.uniform(), size=30, dims=3, form="randomized")
    from gp_emulator_amd.lhd import _crowding
    assert _crowding(a) <= _crowding(b)        # the search keeps the best shuffle, the first included


def test_k_fold_matches_reference():
    g = load_golden("design")
    folds = list(k_fold_cross_validation(range(10), 3))
    assert len(folds) == 3
    for k, (tr, va) in enumerate(folds):
        assert np.array_equal(np.array(tr), g["kfold_train_%d" % k])
        assert np.array_equal(np.array(va), g["kfold_valid_%d" % k])
    tr, va = next(k_fold_cross_validation(list("abcdefg"), 7, randomise=True))
    assert len(va) == 1 and sorted(tr + va) == list("abcdefg")


def _small_gp(seed=0):
    rs = np.random.RandomState(seed)
    x = rs.random_sample((30, 3))
    t = np.sin(x.sum(1))
    gp = GaussianProcess(x, t)
    gp._set_params(np.array([0.1, -0.2, 0.3, 0.0, -5.0]))
    return gp


def test_storage_round_trip_scalar_and_multivariate(tmp_path):
    store = EmulatorStorage(str(tmp_path / "forest.dat"))
    with pytest.raises(IOError):
        store.get_keys()
    gp = _small_gp()
    store.dump_emulator(gp, ("sza", 30, "vza", 0))
    rs = np.random.RandomState(1)
    y = rs.random_sample((25, 2))
    X = np.stack([np.sin(3 * y[:, 0] + w) * np.cos(y[:, 1] * w) for w in np.linspace(0, 2, 40)], axis=1)
    sv = np.linalg.svd(X, compute_uv=False)
    n_pcs = int(np.sum(sv.cumsum() / sv.sum() <= 0.9999))
    assert n_pcs >= 2
    hp = np.tile(np.array([0.0, 0.0, 0.0, -6.0])[:, None], (1, n_pcs))
    mv = MultivariateEmulator(X=X, y=y, hyperparams=hp, thresh=0.9999)
    store.dump_emulator(mv, "spectra")
    assert os.path.exists(store.fname) and not os.path.exists(store.fname + ".npz")
    assert store.get_keys() == [repr(("sza", 30, "vza", 0)), "spectra"]

    back = store.get_emulator(["sza", 30, "vza", 0])          # list or tuple: same key
    assert np.array_equal(back.theta, gp.theta) and np.array_equal(back.invQ, gp.invQ)
    probe = np.random.RandomState(2).random_sample((7, 3))
    for a, b in zip(back.predict(probe), gp.predict(probe)):
        assert np.array_equal(a, b)

    mv2 = store.get_emulator("spectra")
    assert mv2.n_pcs == mv.n_pcs and np.array_equal(mv2.basis_functions, mv.basis_functions)
    f1, j1 = mv.predict(y[3])
    f2, j2 = mv2.predict(y[3])
    assert np.array_equal(f1, f2) and np.array_equal(j1, j2)

    store.dump_emulator(_small_gp(seed=5), ("sza", 30, "vza", 0))   # replace in place
    assert store.get_keys() == [repr(("sza", 30, "vza", 0)), "spectra"]
    assert not np.array_equal(store.get_emulator(("sza", 30, "vza", 0)).inputs, gp.inputs)
    with pytest.raises(KeyError):
        store.get_emulator("missing")
    with pytest.raises(TypeError):
        store.dump_emulator(object(), "x")
    with np.load(store.fname, allow_pickle=False) as f:             # plain arrays only
        assert all(f[k].dtype != object for k in f.files)


def test_multivariate_trains_when_no_hyperparams():
    """reference multivariate_gp.py:180-184: hyperparams=None -> learn_hyperparameters per PC."""
    rs = np.random.RandomState(4)
    y = rs.random_sample((20, 2))
    X = np.stack([np.sin(2 * y[:, 0] + w) + 0.5 * y[:, 1] * w for w in np.linspace(0, 1, 12)], axis=1)
    X = X + 0.01 * rs.standard_normal(X.shape)
    np.random.seed(0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mv = MultivariateEmulator(X=X, y=y, thresh=0.9, n_tries=1)
    assert mv.hyperparams.shape == (4, mv.n_pcs) and mv.n_pcs >= 1
    fwd = mv.predict(y[5], do_deriv=False)
    truncated = mv.compress(X)[:, 5] @ mv.basis_functions     # what the kept PCs can represent
    assert np.max(np.abs(fwd - truncated)) < 0.05
