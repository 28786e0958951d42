"""
oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's angular pair-counting path, used only as a checker by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.  The product
package ``yet_another_wizz_amd`` never imports this module.

Every function cites the reference lines it restates (paths relative to /root/reference/src/yaw).
The O(N1*N2) arithmetic is in ``paircount_oracle.c`` (plain C, -ffp-contract=off); a pure-numpy
version of the same predicate is kept for cross-checking small cases.

Pinned against: tests/golden/*.npz, produced by tools/make_golden.py by running the reference
itself in the build container (see tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc). Building the checker is not using it."""
    so = os.path.join(_HERE, "libyaworacle.so")
    src = os.path.join(_HERE, "paircount_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libyaworacle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libyaworacle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.yaw_oracle_count_tree.restype = None
        _LIB.yaw_oracle_count_jobs.restype = ctypes.c_int
        _LIB.yaw_oracle_max_threads.restype = ctypes.c_int
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


# ----------------------------------------------------------------------------- geometry
def to_3d(ra, dec):
    """coordinates.py:134-147 AngularCoordinates.to_3d (radian in, unit vectors out)."""
    ra = np.asarray(ra, dtype=np.float64)
    dec = np.asarray(dec, dtype=np.float64)
    cos_dec = np.cos(dec)
    return np.cos(ra) * cos_dec, np.sin(ra) * cos_dec, np.sin(dec)


def from_3d(x, y, z):
    """coordinates.py:110-132 AngularCoordinates.from_3d."""
    x, y, z = (np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (x, y, z))
    r_d2 = np.sqrt(x * x + y * y)
    r_d3 = np.sqrt(x * x + y * y + z * z)
    x_normed = np.ones_like(x)
    np.divide(x, r_d2, where=r_d2 > 0.0, out=x_normed)
    sgn = np.where(y == 0.0, 1.0, np.sign(y))  # coordinates.py sgn(): sign with sgn(0) = +1
    ra = np.arccos(x_normed) * sgn % (2.0 * np.pi)
    dec = np.arcsin(z / r_d3)
    return ra, dec


def chord_to_angle(d):
    """coordinates.py:245-268 AngularDistances.from_3d."""
    return 2.0 * np.arcsin(np.asarray(d) / 2.0)


def angular_distance(ra1, dec1, ra2, dec2):
    """coordinates.py:183-204 AngularCoordinates.distance."""
    x1, y1, z1 = to_3d(ra1, dec1)
    x2, y2, z2 = to_3d(ra2, dec2)
    sq = np.stack([(x1 - x2) ** 2, (y1 - y2) ** 2, (z1 - z2) ** 2], axis=-1)
    return chord_to_angle(np.sqrt(sq.sum(axis=-1)))


# ----------------------------------------------------------------------------- K1-K3, K8-K10
def parse_ang_limits(ang_min, ang_max):
    """trees.py:46-81."""
    ang_min = np.atleast_1d(ang_min).astype(np.float64)
    ang_max = np.atleast_1d(ang_max).astype(np.float64)
    if ang_min.ndim != 1 or ang_max.ndim != 1:
        raise ValueError("'ang_min' and 'ang_max' must be 1-dim")
    if len(ang_min) != len(ang_max):
        raise ValueError("length of 'ang_min' and 'ang_max' does not match")
    if np.any(ang_min >= ang_max):
        raise ValueError("'ang_min' < 'ang_max' not satisfied")
    lim = np.column_stack((ang_min, ang_max))
    if np.any(lim < 0.0) or np.any(lim > np.pi):
        raise ValueError("'ang_min' and 'ang_max' not in range [0.0, pi]")
    return lim


def ang_bins_for(ang_limits, rweight, resolution):
    """trees.py:84-117 get_ang_bins: edges pass through 10 ** unique(log10(.))."""
    with np.errstate(divide="ignore"):
        log_range = np.log10(ang_limits)
    if rweight is not None:
        log_bins = np.linspace(log_range.min(), log_range.max(), resolution + 1)
        log_bins = np.concatenate([log_bins, log_range.flatten()])
    else:
        log_bins = log_range.flatten()
    return 10.0 ** np.sort(np.unique(log_bins))


def thresholds_for(ang_bins):
    """coordinates.py:270-277 (r = 2 sin(theta/2)) then libm pow(r, 2.0): SURVEY.md 8(a11)."""
    r = 2.0 * np.sin(np.asarray(ang_bins, dtype=np.float64) / 2.0)
    return np.array([math.pow(float(v), 2.0) for v in r], dtype=np.float64)


def finalize(fine, ang_bins, ang_limits, rweight):
    """trees.py:358-362 + :120-124 + :134-160 (rweight scaling, then per-scale recombination)."""
    counts = np.array(fine, dtype=np.float64)
    if rweight is not None:
        log_edges = np.log10(ang_bins)
        mids = 10.0 ** ((log_edges[:-1] + log_edges[1:]) / 2.0)
        ang_weights = mids**rweight
        counts *= ang_weights / ang_weights.sum()
    out = np.empty(len(ang_limits), dtype=np.float64)
    for i, (lo, hi) in enumerate(ang_limits):
        i0 = np.argmin(np.abs(ang_bins - lo))
        i1 = np.argmin(np.abs(ang_bins - hi))
        out[i] = counts[i0:i1].sum()
    return out


# ----------------------------------------------------------------------------- brute force
def count_tree(xyz1, w1, xyz2, w2, t):
    """C brute force: (int64 counts[E-1], f64 sums[E-1]) for one tree pair (trees.py:348-356)."""
    x1, y1, z1 = (_c(v) for v in xyz1)
    x2, y2, z2 = (_c(v) for v in xyz2)
    w1c, w2c = _c(w1), _c(w2)
    t = _c(t)
    nf = len(t) - 1
    counts = np.zeros(max(nf, 0), dtype=np.int64)
    sums = np.zeros(max(nf, 0), dtype=np.float64)
    lib().yaw_oracle_count_tree(
        ctypes.c_int64(len(x1)), _d(x1), _d(y1), _d(z1), _d(w1c),
        ctypes.c_int64(len(x2)), _d(x2), _d(y2), _d(z2), _d(w2c),
        ctypes.c_int(len(t)), _d(t), counts.ctypes.data_as(_ip), _d(sums),
    )
    return counts, sums


def count_tree_numpy(xyz1, w1, xyz2, w2, t, chunk=512):
    """Same predicate in numpy (separately rounded products and sums) -- small inputs only."""
    x1, y1, z1 = (_c(v) for v in xyz1)
    x2, y2, z2 = (_c(v) for v in xyz2)
    t = np.asarray(t, dtype=np.float64)
    nf = len(t) - 1
    counts = np.zeros(nf, dtype=np.int64)
    sums = np.zeros(nf, dtype=np.float64)
    wb = np.ones(len(x2)) if w2 is None else np.asarray(w2, dtype=np.float64)
    for a0 in range(0, len(x1), chunk):
        sl = slice(a0, a0 + chunk)
        dx = x1[sl, None] - x2[None, :]
        dy = y1[sl, None] - y2[None, :]
        dz = z1[sl, None] - z2[None, :]
        s = (dx * dx + dy * dy) + dz * dz
        wa = np.ones(dx.shape[0]) if w1 is None else np.asarray(w1, dtype=np.float64)[sl]
        ww = wa[:, None] * wb[None, :]
        for j in range(nf):
            m = (s > t[j]) & (s <= t[j + 1])
            counts[j] += int(m.sum())
            sums[j] += float(ww[m].sum())
    return counts, sums


def angular_tree_count(xyz1, w1, xyz2, w2, ang_min, ang_max, rweight=None, resolution=50):
    """Full restatement of AngularTree.count (trees.py:303-362) -> f64[S]."""
    lim = parse_ang_limits(ang_min, ang_max)
    ang_bins = ang_bins_for(lim, rweight, resolution)
    if len(xyz1[0]) == 0 or len(xyz2[0]) == 0:
        return np.zeros(len(lim))
    t = thresholds_for(ang_bins)
    counts, sums = count_tree(xyz1, w1, xyz2, w2, t)
    fine = counts.astype(np.float64) if (w1 is None and w2 is None) else sums
    return finalize(fine, ang_bins, lim, rweight)


def count_jobs(cat1, cat2, jobs, t, threads=None):
    """Job-level C brute force. cat = dict(x, y, z, w|None, nb, off[int64 P*nb+1]);
    returns (int64 counts[n_jobs,B,E-1], f64 sums[n_jobs,B,E-1]). ``threads`` OpenMP threads
    (default: num_threads()); the result does not depend on it."""
    t = _c(t)
    n_bins, n_edges = t.shape
    jobs = np.ascontiguousarray(jobs, dtype=np.int32).reshape(-1, 2)
    nf = n_edges - 1
    counts = np.zeros((len(jobs), n_bins, nf), dtype=np.int64)
    sums = np.zeros((len(jobs), n_bins, nf), dtype=np.float64)
    a = {k: _c(cat1[k]) for k in ("x", "y", "z", "w")}
    b = {k: _c(cat2[k]) for k in ("x", "y", "z", "w")}
    off1 = np.ascontiguousarray(cat1["off"], dtype=np.int64)
    off2 = np.ascontiguousarray(cat2["off"], dtype=np.int64)
    rc = lib().yaw_oracle_count_jobs(
        _d(a["x"]), _d(a["y"]), _d(a["z"]), _d(a["w"]), ctypes.c_int(cat1["nb"]), off1.ctypes.data_as(_ip),
        _d(b["x"]), _d(b["y"]), _d(b["z"]), _d(b["w"]), ctypes.c_int(cat2["nb"]), off2.ctypes.data_as(_ip),
        ctypes.c_int(len(jobs)), jobs.ctypes.data_as(_i32p), ctypes.c_int(n_bins), ctypes.c_int(n_edges),
        _d(t), ctypes.c_int(threads or num_threads()), counts.ctypes.data_as(_ip), _d(sums),
    )
    if rc != 0:
        raise MemoryError("oracle: allocation failed")
    return counts, sums


def num_threads() -> int:
    """Threads the job-level oracle uses: YAW_ORACLE_THREADS, else min(OpenMP max, 16) -- 16 is the
    CPU share of a one-GPU box."""
    env = os.environ.get("YAW_ORACLE_THREADS")
    if env:
        return max(1, int(env))
    return max(1, min(int(lib().yaw_oracle_max_threads()), 16))


# ----------------------------------------------------------------------------- catalogue level
def bin_index(z, edges, closed):
    """trees.py:408-414: np.digitize(z, edges, right=closed=='right'), keep 1..B -> 0..B-1, else -1."""
    idx = np.digitize(z, edges, right=(closed == "right"))
    nb = len(edges) - 1
    return np.where((idx > 0) & (idx <= nb), idx - 1, -1)


def sort_catalog(ra, dec, z, w, patch, n_patches, edges=None, closed="right"):
    """Arrange one catalogue as the job-level oracle wants it: SoA sorted by (patch, bin) + CSR.
    Objects outside the binning are dropped (trees.py:414). ra/dec in radian."""
    x, y, zz = to_3d(ra, dec)
    patch = np.asarray(patch, dtype=np.int64)
    if edges is None:
        nb, k = 1, np.zeros(len(patch), dtype=np.int64)
    else:
        nb, k = len(edges) - 1, bin_index(z, np.asarray(edges, dtype=np.float64), closed)
    keep = k >= 0
    key = patch[keep] * nb + k[keep]
    order = np.argsort(key, kind="stable")
    sel = np.flatnonzero(keep)[order]
    off = np.zeros(n_patches * nb + 1, dtype=np.int64)
    np.cumsum(np.bincount(key, minlength=n_patches * nb), out=off[1:])
    return dict(x=x[sel], y=y[sel], z=zz[sel], w=None if w is None else np.asarray(w, dtype=np.float64)[sel],
                nb=nb, off=off)


def segment_sum_weights(cat, n_patches, n_bins):
    """measurements.py:123-124 + trees.py:225-234,249-258: per (bin, patch) sum of weights
    (float(N) if unweighted); an unbinned catalogue repeats its single tree for every bin."""
    nb, off = cat["nb"], cat["off"]
    out = np.zeros((n_bins, n_patches))
    for p in range(n_patches):
        for k in range(n_bins):
            kk = 0 if nb == 1 else k
            a0, a1 = off[p * nb + kk], off[p * nb + kk + 1]
            out[k, p] = float(a1 - a0) if cat["w"] is None else float(cat["w"][a0:a1].sum())
    return out


def count_pairs(cat1, cat2, jobs, ang_min_per_bin, ang_max_per_bin, n_patches, *, auto, rweight=None,
                resolution=50):
    """measurements.py:307-367 PatchLinkage.count_pairs for a given job list.
    ang_min_per_bin/ang_max_per_bin: f64[B,S] radian (cosmology.py:158-175 evaluated at zmids).
    Returns counts f64[S,B,P,P], sum_weights1 f64[B,P], sum_weights2 f64[B,P]."""
    ang_min_per_bin = np.atleast_2d(ang_min_per_bin)
    ang_max_per_bin = np.atleast_2d(ang_max_per_bin)
    n_bins, n_scales = ang_min_per_bin.shape
    lims = [parse_ang_limits(ang_min_per_bin[k], ang_max_per_bin[k]) for k in range(n_bins)]
    bins = [ang_bins_for(lims[k], rweight, resolution) for k in range(n_bins)]
    n_edges = len(bins[0])
    assert all(len(b) == n_edges for b in bins)
    t = np.stack([thresholds_for(b) for b in bins])
    jobs = np.asarray(jobs, dtype=np.int32).reshape(-1, 2)
    icounts, sums = count_jobs(cat1, cat2, jobs, t)
    weighted = cat1["w"] is not None or cat2["w"] is not None
    fine = sums if weighted else icounts.astype(np.float64)
    out = np.zeros((n_scales, n_bins, n_patches, n_patches))
    for j, (p, q) in enumerate(jobs):
        for k in range(n_bins):
            vals = finalize(fine[j, k], bins[k], lims[k], rweight)
            if auto and p == q:
                vals = vals * 0.5  # measurements.py:362-363
            out[:, k, p, q] = vals
    sw1 = segment_sum_weights(cat1, n_patches, n_bins)
    sw2 = segment_sum_weights(cat2, n_patches, n_bins)
    # measurements.py:358-359 only fills columns of patches that appear in a job
    m1 = np.zeros(n_patches, dtype=bool)
    m2 = np.zeros(n_patches, dtype=bool)
    m1[jobs[:, 0]] = True
    m2[jobs[:, 1]] = True
    sw1[:, ~m1] = 0.0
    sw2[:, ~m2] = 0.0
    return out, sw1, sw2
