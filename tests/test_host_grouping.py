"""yawhip_host_group_columns (the ingest path's stable grouping by patch / by (patch, bin)) against numpy's stable
argsort + gathers -- what the reference does with groupby() (catalog/catalog.py:293, catalog/trees.py:413)."""
import numpy as np
import pytest


def _expected(keys, columns):
    keep = np.flatnonzero(keys >= 0)
    order = keep[np.argsort(keys[keep], kind="stable")]
    return [c[order] for c in columns]


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("n,groups,threads", [(0, 3, 0), (1, 1, 0), (1000, 7, 1), (300_000, 64, 4), (300_001, 1920, 0)])
def test_group_columns_matches_stable_argsort(dtype, n, groups, threads):
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(n + groups)
    keys = rng.integers(-1, groups, n).astype(dtype)  # -1: dropped (objects outside the binning)
    columns = [rng.random(n) for _ in range(4)]
    columns[1] = np.arange(n, dtype=np.float64)  # input position: shows the order inside a group
    outs, sizes = _lib.group_columns(keys, groups, columns, n_threads=threads)
    assert np.array_equal(sizes, np.bincount(keys[keys >= 0], minlength=groups))
    for got, exp in zip(outs, _expected(keys, columns)):
        assert np.array_equal(got, exp)


def test_group_columns_errors():
    from yet_another_wizz_amd import _lib

    with pytest.raises(_lib.YawhipError, match="key >= num_groups"):
        _lib.group_columns(np.array([0, 5, 1]), 3, [np.zeros(3)])
    with pytest.raises(ValueError, match="differ in length"):
        _lib.group_columns(np.array([0, 1]), 3, [np.zeros(3)])


def test_catalog_setup_is_the_same_through_the_library_and_through_numpy(monkeypatch):
    """A catalogue large enough for the library path equals the one built by numpy's stable sort (same object order in
    every patch and in every (patch, bin) segment, same offsets, objects outside the binning dropped)."""
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import catalog

    rng = np.random.default_rng(5)
    n = 250_000
    ra, dec = rng.uniform(0, 2 * np.pi, n), np.arcsin(rng.uniform(-1, 1, n))
    z, w = rng.uniform(0.0, 1.2, n), rng.uniform(0.5, 1.5, n)
    patch = rng.integers(0, 12, n)
    edges = np.linspace(0.1, 1.0, 8)

    def build():
        cat = yaw.Catalog.from_arrays(ra, dec, redshifts=z, weights=w, patch_ids=patch, degrees=False)
        return cat, cat.build_trees(edges), cat.build_trees(None)

    assert n >= catalog.HOST_GROUP_MIN
    a, la, ua = build()
    monkeypatch.setattr(catalog, "HOST_GROUP_MIN", 10**12)
    b, lb, ub = build()
    for name in ("_ra", "_dec", "_w", "_z", "_patch_off"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    for got, exp in zip(a._xyz, b._xyz):
        assert np.array_equal(got, exp)
    for l1, l2 in ((la, lb), (ua, ub)):
        assert np.array_equal(l1.offsets, l2.offsets)
        for name in ("x", "y", "z", "w", "sum_weights"):
            assert np.array_equal(getattr(l1, name), getattr(l2, name)), name
    assert la.num_records < n  # some redshifts lie outside the binning


def test_scatter_rows_matches_numpy():
    """yawhip_host_scatter_rows (dense [S, B, P, P] tensor from per-job values) against zeros + fancy assignment, for
    contiguous values and for the transposed view the job-major device result arrives as."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(3)
    S, B, P, J = 2, 5, 7, 19
    cols = np.sort(rng.choice(P * P, J, replace=False)).astype(np.int64)
    factor = rng.choice([0.5, 1.0], J)
    for vals in (rng.random((S, B, J)), np.moveaxis(rng.random((J, B, 1)), 0, -1)[:, 0][np.newaxis]):
        s_ = vals.shape[0]
        for f in (None, factor):
            exp = np.zeros((s_ * B, P * P))
            exp[:, cols] = vals.reshape(s_ * B, J) * (1.0 if f is None else f)
            got = _lib.scatter_rows((s_, B, P, P), cols, vals, f)
            assert got.shape == (s_, B, P, P) and np.array_equal(got.reshape(s_ * B, P * P), exp)
    assert not _lib.scatter_rows((1, 2, 3, 3), np.zeros(0, dtype=np.int64), np.zeros((1, 2, 0))).any()
    with pytest.raises(_lib.YawhipError, match="out of range"):
        _lib.scatter_rows((1, 1, 2, 2), np.array([4], dtype=np.int64), np.ones((1, 1, 1)))
