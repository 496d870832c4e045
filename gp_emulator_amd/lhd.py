"""Latin-hypercube sampling designs over ``scipy.stats`` distributions.

Python-3 counterpart of gp_emulator/lhd.py:11-269 (the
design utility that produces the training inputs of an emulator; host-only).  Same
signature and return value as the reference's ``lhd``:

    lhd(dist=None, size=None, dims=1, form='randomized', iterations=100,
        showcorrelations=False)  ->  (size, n_vars) array, or None without dist / size

and the same use of ``numpy.random``'s global stream, draw for draw, for
``form='randomized'``: with the same ``numpy.random.seed`` the design equals the reference's
bit for bit (tests/golden/design.npz).  One deliberate difference, in ``form='spacefilling'``:
the reference's search never updates its running best (lhd.py:206-216, ``best`` stays 1e8), so it
returns whichever shuffle came last; here the shuffle with the smallest ``sum 1/d_ij^2`` is kept,
which is what its docstring describes.  ``form='orthogonal'`` raises ``NotImplementedError`` as it
does there (:241-242).
"""
import numpy as np

__all__ = ["lhd"]


def _stratified_unit_samples(n_samples, n_vars):
    """One uniform draw inside each of ``n_samples`` equal strata of [0, 1), per variable
    (reference ``_lhs``, :98-165, on the unit cube).  Draw order: variable-major."""
    u = np.random.random((n_vars, n_samples))          # same stream as n_vars*n_samples scalar draws
    width = 1.0 / n_samples                            # rounded as the reference rounds it
    lower = np.arange(n_samples, dtype=np.float64) * width
    return (lower[None, :] + u * width).T.copy()


def _shuffle_within_columns(design):
    """The reference's ``_mix`` (:167-181): walk down each column, swapping every entry with a
    uniformly drawn row of the same column.  (Not a Fisher-Yates shuffle; kept as is so seeded
    designs match.)"""
    out = np.array(design, copy=True)
    n_rows, n_cols = out.shape
    for col in range(n_cols):
        for row in range(n_rows):
            other = np.random.randint(n_rows)
            out[row, col], out[other, col] = out[other, col], out[row, col]
    return out


def _crowding(points):
    """sum over pairs of 1 / squared distance: small when the points are spread out (:196-204)."""
    diff = points[:, None, :] - points[None, :, :]
    d2 = np.einsum("ijk,ijk->ij", diff, diff)
    iu = np.triu_indices(points.shape[0], k=1)
    with np.errstate(divide="ignore"):
        return float(np.sum(1.0 / d2[iu]))


def _correlation_report(design):
    """Pairwise correlation matrix, its pseudo-inverse and the variance inflation factor
    (:244-262)."""
    centred = design - design.mean(axis=0)
    norm = np.sqrt(np.sum(centred ** 2, axis=0))
    cor = (centred.T @ centred) / np.outer(norm, norm)
    inv = np.linalg.pinv(cor)
    return cor, inv, float(np.max(np.diag(inv)))


def lhd(dist=None, size=None, dims=1, form="randomized", iterations=100, showcorrelations=False):
    """Latin-hypercube design: ``size`` rows, one column per distribution in ``dist`` (or
    ``dims`` columns of the single frozen distribution ``dist``), each column holding exactly
    one sample from each of ``size`` equal-probability strata of its distribution."""
    assert dims > 0, 'kwarg "dims" must be at least 1'
    if not size or not dist:
        return None
    several = hasattr(dist, "__getitem__")
    n_vars = len(dist) if several else dims

    if form == "randomized":
        unit = _shuffle_within_columns(_stratified_unit_samples(size, n_vars))
    elif form == "spacefilling":
        unit = _shuffle_within_columns(_stratified_unit_samples(size, n_vars))
        best, best_score = unit, _crowding(unit)
        for _ in range(max(int(iterations) - 1, 0)):
            unit = _shuffle_within_columns(unit)
            score = _crowding(unit)
            if score < best_score:
                best, best_score = unit, score
        unit = best
    elif form == "orthogonal":
        raise NotImplementedError("Sorry. The orthogonal space-filling algorithm hasn't been "
                                  "implemented yet.")
    else:
        raise ValueError('Invalid "form" value: %s' % (form))

    design = np.empty_like(unit)
    for i in range(n_vars):
        design[:, i] = (dist[i] if several else dist).ppf(unit[:, i])

    if design.shape[1] > 1 and showcorrelations:
        cor, inv, vif = _correlation_report(design)
        print("Correlation Matrix:\n", cor)
        print("Inverted Correlation Matrix:\n", inv)
        print("Variance Inflation Factor (VIF):", vif)
    return design
