"""GPU: randomised mid-size set-ups (2e4 - 3e5 objects a side) -- every culling path against the brute-force FP64 kernel
of the same library (itself held to the oracle and the reference's vectors in test_gpu_kernel_parity.py), over random
footprints, clumpiness, patch counts, binnings, scales, weights, strip grids, sort axes and kernel tunables. The sizes
reach what the small oracle-checked cases cannot: windows of several LDS stages, lane tiles with ragged ends in every
run, thousands of jobs, counters beyond 2^32 are covered elsewhere."""
import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from yet_another_wizz_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


def _sample(rng, n, cap, clumps, frac, P, nb, weighted, centres):
    n_cl = int(n * frac)
    which = rng.integers(0, len(clumps[0]), n_cl)
    v = np.concatenate([clumps[0][which] + clumps[1][which, None] * rng.normal(size=(n_cl, 3)),
                        centres[0] + np.tan(min(cap, 1.4)) * 0.6 * rng.normal(size=(n - n_cl, 3))])
    v /= np.linalg.norm(v, axis=1)[:, None]
    ra, dec = np.arctan2(v[:, 1], v[:, 0]) % (2 * np.pi), np.arcsin(np.clip(v[:, 2], -1, 1))
    # nearest of P random directions of the footprint
    patch = np.argmax(v @ centres[1].T, axis=1)
    patch[:P] = np.arange(P)  # no empty patch
    z = np.concatenate([clumps[2][which] + 0.02 * rng.normal(size=n_cl), rng.uniform(0.0, 1.0, n - n_cl)])
    w = rng.uniform(0.2, 3.0, n) if weighted else None
    edges = np.linspace(0.05, 0.95, nb + 1) if nb > 1 else None
    return oracle.sort_catalog(ra, dec, z, w, patch, P, edges, "right")


@pytest.mark.parametrize("seed", range(40))
def test_culling_paths_equal_brute_force(ctx, seed):
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(7000 + seed)
    P = int(rng.choice([3, 7, 20, 39, 64]))
    B = int(rng.choice([1, 3, 12, 40, 150]))
    n1, n2 = int(rng.integers(20_000, 300_000)), int(rng.integers(20_000, 300_000))
    cap = float(rng.choice([0.03, 0.15, 0.6]))  # footprint (rad)
    centre = rng.normal(size=3); centre /= np.linalg.norm(centre)
    n_clumps = int(rng.integers(5, 60))
    c_dir = centre + np.tan(cap) * 0.5 * rng.normal(size=(n_clumps, 3))
    c_dir /= np.linalg.norm(c_dir, axis=1)[:, None]
    clumps = (c_dir, cap * 10.0 ** rng.uniform(-3.0, -1.0, n_clumps), rng.uniform(0.1, 0.9, n_clumps))
    p_dir = centre + np.tan(cap) * 0.5 * rng.normal(size=(P, 3))
    p_dir /= np.linalg.norm(p_dir, axis=1)[:, None]
    frac = float(rng.choice([0.0, 0.5, 0.9]))
    w1, w2 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    nb2 = B if rng.integers(0, 2) else 1
    c1 = _sample(rng, n1, cap, clumps, frac, P, B, w1, (centre, p_dir))
    c2 = _sample(rng, n2, cap, clumps, frac, P, nb2, w2, (centre, p_dir))
    n_scales = int(rng.integers(1, 4))
    lo = np.sort(rng.uniform(0.002, 0.02, n_scales)) * cap
    hi = lo * rng.uniform(2.0, 8.0, n_scales)
    # one in three: separation weights, i.e. dozens of fine bins per z-bin (edge table in LDS, fine bin by binary search)
    rw, res = (-1.0, int(rng.integers(5, 60))) if rng.integers(0, 3) == 0 else (None, None)
    t_row = oracle.thresholds_for(oracle.ang_bins_for(oracle.parse_ang_limits(lo, hi), rw, res))
    t = np.stack([t_row] * B) if rng.integers(0, 2) else np.stack(
        [oracle.thresholds_for(oracle.ang_bins_for(oracle.parse_ang_limits(lo * (1 + 0.05 * k), hi * (1 + 0.05 * k)), rw, res))
         for k in range(B)])
    jobs = np.array([(p, q) for p in range(P) for q in range(P) if rng.random() < 0.7 or p == q], dtype=np.int32)
    micro = int(max(1000, np.ceil(1.02e6 * np.sqrt(t.max()) / 50) * 50))  # just above the largest chord: three partner strips
    # also: no strips; narrower strips (five to nine partner strips, several windows groups per tile); wider ones
    micro = int(rng.choice([micro, micro, 0, max(1000, int(micro / 2.7)), max(1000, int(micro / 3.9)), min(3 * micro, 2000000)]))
    axis = int(rng.integers(0, 3))
    try:
        ctx.set_option("strip_width_micro", micro)
        up = lambda c: _lib.DeviceCatalog(ctx, c["x"], c["y"], c["z"], c["w"], P, c["nb"], c["off"], sort_axis=axis)
        d1, d2 = up(c1), up(c2)
        for da, db in ((d1, d2), (d1, d1)):
            ctx.set_option("tile_r", 0); ctx.set_option("band_cap", 0); ctx.set_option("hist_copies_log2", -1)
            exp_c, exp_s, st = _lib.count_pairs(ctx, da, db, jobs, t, kernel="exact", want_counts=True, want_sums=True)
            assert exp_c.sum() > 0
            weighted = da.weighted or db.weighted
            for kernel in ("band", "sweep", "auto"):
                ctx.set_option("tile_r", int(rng.choice([0, 1, 2, 4])))
                ctx.set_option("band_cap", int(rng.choice([0, 192, 288])))
                ctx.set_option("hist_copies_log2", int(rng.choice([-1, 0, 3])))
                ctx.set_option("band_batch_log2", int(rng.choice([-1, 0, 2, 4])))
                counts, sums, stats = _lib.count_pairs(ctx, da, db, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
                assert np.array_equal(counts, exp_c), (seed, kernel)
                if weighted:
                    np.testing.assert_allclose(sums, exp_s, rtol=1e-11, atol=0)
                else:
                    assert np.array_equal(sums, exp_c.astype(np.float64))
                assert stats.candidate_pairs == st.candidate_pairs
    finally:
        for key, val in (("strip_width_micro", _lib.DEFAULT_STRIP_MICRO), ("tile_r", 0), ("band_cap", 0), ("hist_copies_log2", -1),
                         ("band_batch_log2", -1)):
            ctx.set_option(key, val)
