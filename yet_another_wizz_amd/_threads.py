"""Host-side helpers that run numpy's elementwise kernels on slices of large columns in a small thread pool.

Catalogue preparation (RA/Dec -> unit vectors, redshift binning, regrouping of columns) is O(N) numpy work that has to
stay on the host -- the unit vectors must be numpy's cos / sin values, the pair predicate runs on exactly these -- but
numpy releases the GIL inside its loops, so slices can be processed side by side. Every function here is elementwise
(or a pure gather): the result is the same array an unsliced call returns (numpy handles loop tails with masked vector
operations; ``self_check`` verifies that once per process on the machine at hand and turns the slicing off if not).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

__all__ = ["map_slices", "take", "radec_to_xyz", "digitize", "pool_size"]

MIN_PARALLEL = 1 << 20  # below this a single numpy call is as fast
_pool = None
_checked = None


def pool_size() -> int:
    return max(1, min(16, (os.cpu_count() or 1)))


def _executor():
    global _pool
    if _pool is None:
        _pool = ThreadPoolExecutor(max_workers=pool_size(), thread_name_prefix="yaw-host")
    return _pool


def _slices(n: int):
    parts = pool_size() * 2
    step = max(1 << 16, -(-n // parts))
    return [slice(lo, min(lo + step, n)) for lo in range(0, n, step)]


def self_check() -> bool:
    """cos / sin of a slice equal the slice of cos / sin of the whole (at odd offsets and lengths)."""
    global _checked
    if _checked is None:
        rng = np.random.default_rng(20240607)
        a = rng.uniform(-7.0, 7.0, 100_003)
        whole = (np.cos(a), np.sin(a))
        ok = True
        for lo, n in ((1, 65_537), (3, 17), (12_345, 50_001), (99_990, 13)):
            ok &= np.array_equal(np.cos(a[lo : lo + n]), whole[0][lo : lo + n]) and np.array_equal(np.sin(a[lo : lo + n]), whole[1][lo : lo + n])
        _checked = bool(ok)
    return _checked


def map_slices(fn, n: int, out_dtypes):
    """``fn(slice) -> tuple of arrays`` evaluated over slices of range(n); returns the concatenated outputs
    (pre-allocated, written in place)."""
    outs = tuple(np.empty(n, dtype=dt) for dt in out_dtypes)
    if n < MIN_PARALLEL or pool_size() == 1 or not self_check():
        res = fn(slice(0, n))
        for o, r in zip(outs, res):
            o[:] = r
        return outs

    def work(sl):
        for o, r in zip(outs, fn(sl)):
            o[sl] = r

    list(_executor().map(work, _slices(n)))
    return outs


def radec_to_xyz(ra, dec):
    """x = cos(ra) cos(dec), y = sin(ra) cos(dec), z = sin(dec) (reference coordinates.py:143-146), sliced."""
    ra = np.asarray(ra, dtype=np.float64)
    dec = np.asarray(dec, dtype=np.float64)

    def part(sl):
        r, d = ra[sl], dec[sl]
        cos_dec = np.cos(d)
        return np.cos(r) * cos_dec, np.sin(r) * cos_dec, np.sin(d)

    return map_slices(part, len(ra), (np.float64, np.float64, np.float64))


def take(column, index):
    """``column[index]`` (a gather), sliced over ``index``."""
    column = np.asarray(column)
    index = np.asarray(index)
    (out,) = map_slices(lambda sl: (column[index[sl]],), len(index), (column.dtype,))
    return out


def digitize(values, edges, right: bool):
    """``np.digitize(values, edges, right=right)``, sliced."""
    values = np.asarray(values)
    (out,) = map_slices(lambda sl: (np.digitize(values[sl], edges, right=right),), len(values), (np.int64,))
    return out
