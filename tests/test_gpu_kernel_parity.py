"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against
(1) golden vectors captured from the reference and (2) the CPU oracle on seeded inputs.
Unweighted counts must be bit-identical; weighted sums within 1e-10 relative (north_star)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu
RTOL_W = 1e-10


@pytest.fixture(scope="module")
def ctx():
    from yet_another_wizz_amd import _lib

    c = _lib.Context(0)
    yield c
    c.close()


def _upload(ctx, cat):
    from yet_another_wizz_amd import _lib

    n_patches = (len(cat["off"]) - 1) // cat["nb"]
    return _lib.DeviceCatalog(ctx, cat["x"], cat["y"], cat["z"], cat["w"], n_patches, cat["nb"], cat["off"])


def _single(xyz, w):
    return dict(x=xyz[:, 0].copy(), y=xyz[:, 1].copy(), z=xyz[:, 2].copy(), w=w, nb=1,
                off=np.array([0, len(xyz)], dtype=np.int64))


KERNELS = ["exact", "filter", "sweep", "band"]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("tile_r", [1, 2, 4])
def test_single_job_golden(ctx, kernel, tile_r):
    from yet_another_wizz_amd import _lib

    sj = load_golden("single_job.npz")
    ctx.set_option("tile_r", tile_r)
    cats = {}
    for wname in ("uu", "ww", "wu", "uw"):
        w1 = sj["w1"] if wname[0] == "w" else None
        w2 = sj["w2"] if wname[1] == "w" else None
        cats[wname] = (_upload(ctx, _single(sj["xyz1"], w1)), _upload(ctx, _single(sj["xyz2"], w2)))
    n_checked = 0
    for key in (str(k) for k in sj["case_names"]):
        parts = key.split(".")
        wname, kind = parts[1], parts[-1]
        c1, c2 = cats[wname]
        if kind == "auto":
            c2 = c1
        t = sj[key + ".t"][None, :]
        counts, sums, stats = _lib.count_pairs(ctx, c1, c2, [[0, 0]], t, kernel=kernel, want_counts=True, want_sums=True)
        exp = sj[key + ".fine"]
        if wname == "uu":
            assert np.array_equal(counts[0, 0].astype(np.float64), exp), key
            assert np.array_equal(sums[0, 0], exp), key
        else:
            np.testing.assert_allclose(sums[0, 0], exp, rtol=RTOL_W, atol=0, err_msg=key)
        assert stats.candidate_pairs == c1.n * c2.n
        n_checked += 1
    assert n_checked >= 40
    ctx.set_option("tile_r", 0)


@pytest.mark.parametrize("kernel", KERNELS)
def test_greatcircle_golden(ctx, kernel):
    from yet_another_wizz_amd import _lib

    gc = load_golden("greatcircle.npz")
    DELTA = 1e-9
    w = np.full(len(gc["xyz"]), 2.0)
    tree = _upload(ctx, _single(gc["xyz"], w))
    single = _upload(ctx, _single(gc["single_xyz"], np.array([2.0])))
    for am in (1.0, 2.0, 10.0, 89.0):
        hi = am + DELTA
        lim = oracle.parse_ang_limits(np.deg2rad(hi - 1.0), np.deg2rad(hi))
        ang_bins = oracle.ang_bins_for(lim, None, None)
        t = oracle.thresholds_for(ang_bins)[None, :]
        _, sums, _ = _lib.count_pairs(ctx, tree, single, [[0, 0]], t, kernel=kernel)
        got = oracle.finalize(sums[0, 0], ang_bins, lim, None)
        assert got == 16.0 and np.array_equal(got, gc[f"single_{int(am)}"])
    # 88 one-degree annuli at once (E >= 8)
    hi = np.arange(1.0, 89.0) + DELTA
    lim = oracle.parse_ang_limits(np.deg2rad(hi - 1.0), np.deg2rad(hi))
    ang_bins = oracle.ang_bins_for(lim, None, None)
    t = oracle.thresholds_for(ang_bins)[None, :]
    _, sums, _ = _lib.count_pairs(ctx, tree, single, [[0, 0]], t, kernel=kernel)
    assert np.array_equal(oracle.finalize(sums[0, 0], ang_bins, lim, None), gc["bins_89"])
    # self count: ordered pairs, self pairs excluded (reference test_trees.py:239-247)
    utree = _upload(ctx, _single(gc["xyz"], None))
    lim = oracle.parse_ang_limits(*(np.deg2rad([0.0, 1.0]) + DELTA))
    ang_bins = oracle.ang_bins_for(lim, None, None)
    t = oracle.thresholds_for(ang_bins)[None, :]
    counts, _, _ = _lib.count_pairs(ctx, utree, utree, [[0, 0]], t, kernel=kernel)
    assert counts[0, 0, 0] == 4 * 6 + 2 * (len(gc["xyz"]) - 6) == gc["dualtree"][0]


def _random_catalog(rng, n, n_patches, nb, weighted, dense_box=2.0):
    ra = np.deg2rad(rng.uniform(100.0, 100.0 + dense_box, n))
    dec = np.arcsin(rng.uniform(np.sin(np.deg2rad(30.0)), np.sin(np.deg2rad(30.0 + dense_box)), n))
    patch = rng.integers(0, n_patches, n)
    patch[patch == 2] = 1  # patch 2 stays empty
    z = rng.uniform(0.0, 1.0, n)
    edges = np.linspace(0.1, 0.9, nb + 1) if nb > 1 else None
    w = rng.uniform(0.5, 1.5, n) if weighted else None
    return oracle.sort_catalog(ra, dec, z, w, patch, n_patches, edges, "right")


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("nb2", [1, 5])
@pytest.mark.parametrize("weights", ["uu", "wu", "uw", "ww"])
def test_ragged_jobs_vs_oracle(ctx, kernel, nb2, weights):
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(99 + nb2)
    P, B = 6, 5
    c1 = _random_catalog(rng, 7000, P, B, weights[0] == "w")
    c2 = _random_catalog(rng, 9000, P, nb2, weights[1] == "w")
    jobs = np.array([(p, q) for p in range(P) for q in range(P) if (p + q) % 3 != 1], dtype=np.int32)
    lim = oracle.parse_ang_limits(np.array([0.5, 2.0]) * np.pi / 10800, np.array([3.0, 8.0]) * np.pi / 10800)
    t = np.stack([oracle.thresholds_for(oracle.ang_bins_for(lim * (1.0 + 0.1 * k), None, None)) for k in range(B)])
    d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
    exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
    for tile_r in (0, 1, 4):
        ctx.set_option("tile_r", tile_r)
        counts, sums, stats = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
        assert np.array_equal(counts, exp_c)
        if weights == "uu":
            assert np.array_equal(sums, exp_c.astype(np.float64))
        else:
            np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
        assert exp_c.sum() > 1000
    ctx.set_option("tile_r", 0)


@pytest.mark.parametrize("weights", ["uu", "ww"])
@pytest.mark.parametrize("n1", [300, 1200, 2600])
def test_sparse_items_share_their_bands(ctx, n1, weights):
    """Lane tiles with a handful of objects facing ONE window that fits the stage: k_count_band32 shares the bands of the few
    lanes that have one out over the wave (8, 4 or 2 lanes per lane's objects, every 8th / 4th / 2nd entry each; yawhip.hip,
    YAW_B32_SHARE). Runs of ~15, ~60 and ~130 objects per (patch, strip); one annulus with common and with per-bin edges
    (thresholds per lane object), two scales sharing no edge (four edges, cumulative counters), 1, 2 and 4 objects per lane,
    with and without the strip grid, binned x unbinned and the binned catalogue against itself."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(4242 + n1)
    P, B = 3, 4
    c1 = _random_catalog(rng, n1, P, B, weights[0] == "w")
    c2 = _random_catalog(rng, 5000, P, 1, weights[1] == "w")
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    arcmin = np.pi / 10800
    one = oracle.parse_ang_limits([0.8 * arcmin], [4.0 * arcmin])
    two = oracle.parse_ang_limits(np.array([0.5, 2.5]) * arcmin, np.array([2.0, 5.0]) * arcmin)
    tables = {
        "one annulus": np.tile(oracle.thresholds_for(oracle.ang_bins_for(one, None, None)), (B, 1)),
        "per-bin annulus": np.stack([oracle.thresholds_for(oracle.ang_bins_for(one * (1.0 + 0.07 * k), None, None)) for k in range(B)]),
        "four edges": np.tile(oracle.thresholds_for(oracle.ang_bins_for(two, None, None)), (B, 1)),
    }
    try:
        for strip_micro in (_lib.DEFAULT_STRIP_MICRO, 0):
            ctx.set_option("strip_width_micro", strip_micro)
            d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
            for name, t in tables.items():
                exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
                exp_self, exp_self_s = oracle.count_jobs(c1, c1, jobs, t)
                assert exp_c.sum() > 200
                for tile_r in (0, 1, 4):
                    ctx.set_option("tile_r", tile_r)
                    for (da, db, ec, es) in ((d1, d2, exp_c, exp_s), (d1, d1, exp_self, exp_self_s)):
                        counts, sums, stats = _lib.count_pairs(ctx, da, db, jobs, t, kernel="band", want_counts=True, want_sums=True)
                        assert stats.kernel_used == _lib.KERNEL_BAND, (name, strip_micro)
                        if strip_micro and db is d2:  # (sparse binned x binned counts run on the plain layout: float64 band kernel)
                            assert stats.band_variant == 32, (name, stats.band_variant)
                        assert np.array_equal(counts, ec), (name, strip_micro, tile_r)
                        if weights == "uu":
                            assert np.array_equal(sums, ec.astype(np.float64))
                        else:
                            np.testing.assert_allclose(sums, es, rtol=RTOL_W, atol=0)
    finally:
        ctx.set_option("tile_r", 0)
        ctx.set_option("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)


@pytest.mark.parametrize("weights", ["uu", "ww"])
def test_merged_triple_runs_and_item_segments(ctx, weights):
    """The streamed side of a float32 band count read from MERGED runs of three neighbouring strips (k_merge_triples; one
    window per item) or from the strips themselves (three windows), the item list in eight segments or in one: every
    combination gives the oracle's counts. A strip grid as wide as the largest separation (reach 1) is what merging needs;
    `triple_runs = 2` merges also where the merged window no longer fits the stage (it then goes through it in pieces), with
    the small and the big stage; the dense unbinned side makes windows of several hundred entries."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(777)
    P, B = 4, 3
    c1 = _random_catalog(rng, 9000, P, B, weights[0] == "w", dense_box=3.0)
    c2 = _random_catalog(rng, 60000, P, 1, weights[1] == "w", dense_box=3.0)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    arcmin = np.pi / 10800
    lim = oracle.parse_ang_limits([1.0 * arcmin], [10.0 * arcmin])
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)), (B, 1))
    exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
    exp_self, exp_self_s = oracle.count_jobs(c1, c1, jobs, t)
    micro = int(np.sqrt(t.max()) * 1.02e6) + 1  # just above the largest chord: partners in the strips c - 1, c, c + 1
    seen = set()
    try:
        ctx.set_option("strip_width_micro", micro)
        d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
        for triple in (0, 1, 2):
            ctx.set_option("triple_runs", triple)
            for segments in (0, 1):
                ctx.set_option("item_segments", segments)
                for cap in (0, 320, 512):
                    ctx.set_option("band_cap", cap)
                    for (da, db, ec, es) in ((d1, d2, exp_c, exp_s), (d1, d1, exp_self, exp_self_s)):
                        counts, sums, st = _lib.count_pairs(ctx, da, db, jobs, t, kernel="band", want_counts=True, want_sums=True)
                        assert st.kernel_used == _lib.KERNEL_BAND
                        assert np.array_equal(counts, ec), (triple, segments, cap)
                        if weights == "uu":
                            assert np.array_equal(sums, ec.astype(np.float64))
                        else:
                            np.testing.assert_allclose(sums, es, rtol=RTOL_W, atol=0)
                        if db is d2:
                            assert st.band_variant == 32
                            seen.add((triple, bool(st.merged_triples)))
        assert (0, False) in seen and (2, True) in seen  # both forms of the streamed side ran
    finally:
        for key, value in (("triple_runs", 1), ("item_segments", 1), ("band_cap", 0), ("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)):
            ctx.set_option(key, value)


@pytest.mark.parametrize("kernel", KERNELS)
def test_empty_and_degenerate(ctx, kernel):
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(5)
    c1 = _random_catalog(rng, 300, 3, 1, False)
    empty = dict(x=np.empty(0), y=np.empty(0), z=np.empty(0), w=None, nb=1, off=np.zeros(4, dtype=np.int64))
    d1, de = _upload(ctx, c1), _upload(ctx, empty)
    t = np.array([[1e-8, 1e-6]])
    counts, _, stats = _lib.count_pairs(ctx, d1, de, [[0, 0], [1, 1]], t, kernel=kernel)
    assert counts.sum() == 0 and stats.candidate_pairs == 0
    counts, _, _ = _lib.count_pairs(ctx, d1, d1, np.empty((0, 2), dtype=np.int32), t, kernel=kernel)
    assert counts.shape == (0, 1, 1)
    # many edges -> shared-histogram variant
    edges = np.linspace(1e-9, 1e-5, 300)[None, :]
    counts, _, _ = _lib.count_pairs(ctx, d1, d1, [[1, 1]], edges, kernel=kernel)
    exp, _ = oracle.count_jobs(c1, c1, [[1, 1]], edges)
    assert np.array_equal(counts, exp)


def test_errors(ctx):
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(6)
    c1 = _random_catalog(rng, 100, 3, 1, False)
    c4 = _random_catalog(rng, 100, 4, 1, False)
    d1, d4 = _upload(ctx, c1), _upload(ctx, c4)
    with pytest.raises(_lib.YawhipError, match="patch counts differ"):
        _lib.count_pairs(ctx, d1, d4, [[0, 0]], np.array([[1e-6, 1e-5]]))
    with pytest.raises(_lib.YawhipError, match="ascending"):
        _lib.count_pairs(ctx, d1, d1, [[0, 0]], np.array([[1e-5, 1e-6]]))
    with pytest.raises(_lib.YawhipError, match="patch id"):
        _lib.count_pairs(ctx, d1, d1, [[0, 7]], np.array([[1e-6, 1e-5]]))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("theta_arcmin", [0.05, 1.0, 10.0, 120.0])
def test_borderline_pairs(ctx, kernel, theta_arcmin):
    """Partners placed within 1e-6 ... 1e-16 (relative) of the outer and inner edge: any pre-filter or
    culling step must be conservative, the exact predicate decides. Bit parity with the oracle."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(int(theta_arcmin * 100))
    n = 3000
    ra = rng.uniform(0.0, 2 * np.pi, n)
    dec = np.arcsin(rng.uniform(-1, 1, n))
    a = np.column_stack(oracle.to_3d(ra, dec))
    # a random tangent direction per point
    v = rng.normal(size=(n, 3))
    v -= (v * a).sum(1, keepdims=True) * a
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    theta = theta_arcmin * np.pi / 10800
    lim = oracle.parse_ang_limits([theta * 0.1], [theta])
    ang_bins = oracle.ang_bins_for(lim, None, None)
    t = oracle.thresholds_for(ang_bins)[None, :]
    edge = np.where(rng.random(n) < 0.5, ang_bins[1], ang_bins[0])
    delta = rng.choice([0.0, 1e-16, -1e-16, 3e-16, -3e-16, 1e-14, -1e-14, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6], n)
    ang = edge * (1.0 + delta)
    b = a * np.cos(ang)[:, None] + v * np.sin(ang)[:, None]
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    c1, c2 = _single(a, None), _single(b, None)
    d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
    exp, _ = oracle.count_jobs(c1, c2, [[0, 0]], t)
    counts, _, _ = _lib.count_pairs(ctx, d1, d2, [[0, 0]], t, kernel=kernel)
    assert np.array_equal(counts, exp)
    assert exp.sum() > n // 4  # the engineered partners dominate the count
    # swap roles (lane side <-> streamed side)
    counts, _, _ = _lib.count_pairs(ctx, d2, d1, [[0, 0]], t, kernel=kernel)
    assert np.array_equal(counts, exp)


def test_non_unit_vectors_fall_back_to_exact(ctx):
    """The FP32 pre-filter presumes unit vectors; other inputs must still be counted exactly."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(11)
    a = rng.normal(size=(500, 3)) * 0.01 + np.array([0.3, 0.2, 0.1])
    b = rng.normal(size=(700, 3)) * 0.01 + np.array([0.3, 0.2, 0.1])
    c1, c2 = _single(a, None), _single(b, None)
    t = np.array([[1e-5, 1e-4, 4e-4]])
    exp, _ = oracle.count_jobs(c1, c2, [[0, 0]], t)
    counts, _, stats = _lib.count_pairs(ctx, _upload(ctx, c1), _upload(ctx, c2), [[0, 0]], t, kernel="filter")
    assert np.array_equal(counts, exp) and exp.sum() > 1000
    assert stats.kernel_used == _lib.KERNEL_EXACT


@pytest.mark.parametrize("kernel", ["auto", "band", "sweep"])
def test_non_unit_vectors_beyond_the_band_sentinel(ctx, kernel):
    """The band kernels park finished lanes on a sentinel at coordinate 4.0 and bound their searches by it: input that
    is not made of unit vectors -- here coordinates around 6, beyond the sentinel -- must never reach them: they are
    counted by the exact FP64 brute-force kernel, counts equal the oracle's."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(12)
    centre = np.array([6.0, 5.0, 4.5])
    a = rng.normal(size=(900, 3)) * 0.02 + centre
    b = rng.normal(size=(1100, 3)) * 0.02 + centre
    c1, c2 = _single(a, None), _single(b, None)
    t = np.array([[1e-5, 2e-4, 1e-3]])
    exp, _ = oracle.count_jobs(c1, c2, [[0, 0]], t)
    counts, _, stats = _lib.count_pairs(ctx, _upload(ctx, c1), _upload(ctx, c2), [[0, 0]], t, kernel=kernel)
    assert np.array_equal(counts, exp) and exp.sum() > 1000
    # (a sweep request keeps its windowed items, evaluated by the exact FP64 kernel: no sentinel involved)
    assert stats.kernel_used == (_lib.KERNEL_SWEEP if kernel == "sweep" else _lib.KERNEL_EXACT)


@pytest.mark.parametrize("fp32", [0, 1])
def test_band_kernel_float64_and_float32_classification_agree(ctx, fp32):
    """``band_fp32`` selects between the band kernel that decides every entry in float64 (k_count_band) and the one that
    classifies in float32 and re-evaluates its guard bands in float64 (k_count_band32, the default): identical results
    on merged items (binned x unbinned, per-bin thresholds, 2 and 4 edges) and per-bin items, weighted or not."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(321 + 0)
    P, B = 5, 6
    ctx.set_option("band_fp32", fp32)
    try:
        for weights in ("uu", "wu", "uw", "ww"):
            c1 = _random_catalog(rng, 9000, P, B, weights[0] == "w", dense_box=1.5)
            for nb2 in (1, B):
                c2 = _random_catalog(rng, 11000, P, nb2, weights[1] == "w", dense_box=1.5)
                jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
                for lim in (oracle.parse_ang_limits(np.array([1.0]) * np.pi / 10800, np.array([9.0]) * np.pi / 10800),
                            oracle.parse_ang_limits(np.array([0.5, 2.0]) * np.pi / 10800, np.array([3.0, 8.0]) * np.pi / 10800)):
                    t = np.stack([oracle.thresholds_for(oracle.ang_bins_for(lim * (1.0 + 0.07 * k), None, None)) for k in range(B)])
                    d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
                    exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
                    counts, sums, stats = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel="band", want_counts=True, want_sums=True)
                    assert stats.kernel_used == _lib.KERNEL_BAND
                    assert np.array_equal(counts, exp_c) and exp_c.sum() > 1000
                    if weights == "uu":
                        assert np.array_equal(sums, exp_c.astype(np.float64))
                    else:
                        np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
    finally:
        ctx.set_option("band_fp32", 1)


@pytest.mark.parametrize("weights", ["uu", "ww"])
def test_merged_path_many_edges_and_bins(ctx, weights):
    """Cross-correlation fast path (binned x unbinned) with a fine radial binning (rweight-like, E = 58)
    and thresholds that differ per redshift bin; sweep == exact == oracle."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(2718)
    P, B = 4, 7
    c1 = _random_catalog(rng, 12000, P, B, weights[0] == "w", dense_box=1.2)
    c2 = _random_catalog(rng, 15000, P, 1, weights[1] == "w", dense_box=1.2)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    t = []
    for k in range(B):
        lim = oracle.parse_ang_limits(np.array([0.3, 1.0]) * (1 + 0.2 * k) * np.pi / 10800,
                                      np.array([2.0, 9.0]) * (1 + 0.2 * k) * np.pi / 10800)
        t.append(oracle.thresholds_for(oracle.ang_bins_for(lim, -1.0, 55)))
    t = np.stack(t)
    assert t.shape == (B, 58)
    d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
    exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
    for kernel in ("band", "sweep", "exact"):
        counts, sums, stats = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
        assert np.array_equal(counts, exp_c), kernel
        if weights == "uu":
            assert np.array_equal(sums, exp_c.astype(np.float64))
        else:
            np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
    assert exp_c.sum() > 50000 and (exp_c > 0).sum() > 0.5 * exp_c.size


def test_many_survivors_per_object(ctx):
    """Dense clump: thousands of lane objects within the outer edge of every streamed object, so the
    per-wave survivor queue fills and drains many times per stage."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(31)
    n1, n2 = 900, 5000
    def clump(n, zbins):
        ra = np.deg2rad(200.0 + rng.normal(0, 0.01, n)); dec = np.deg2rad(-45.0 + rng.normal(0, 0.01, n))
        z = rng.uniform(0.1, 0.9, n)
        return oracle.sort_catalog(ra, dec, z, None, np.zeros(n, dtype=int), 1, np.linspace(0.1, 0.9, zbins + 1) if zbins > 1 else None, "right")
    c1, c2 = clump(n1, 3), clump(n2, 1)
    lim = oracle.parse_ang_limits([0.2 * np.pi / 10800], [3.0 * np.pi / 10800])
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)), (3, 1))
    exp, _ = oracle.count_jobs(c1, c2, [[0, 0]], t)
    for kernel in ("band", "sweep", "filter"):
        counts, _, _ = _lib.count_pairs(ctx, _upload(ctx, c1), _upload(ctx, c2), [[0, 0]], t, kernel=kernel)
        assert np.array_equal(counts, exp), kernel
    assert exp.sum() > 0.5 * n1 * n2


def test_sort_axis_variants(ctx):
    """A polar-cap footprint is flat in z; the library can sort along x or y instead. Any axis (and a
    mismatch between the two catalogues, which only disables the culling) gives the same counts."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(77)
    n1, n2, P, B = 20000, 30000, 3, 4

    def polar(n, nb):
        theta = np.sqrt(rng.uniform(0, 1, n)) * np.deg2rad(4.0)  # within 4 degrees of the north pole
        phi = rng.uniform(0, 2 * np.pi, n)
        patch = np.minimum((phi / (2 * np.pi) * P).astype(int), P - 1)
        z = rng.uniform(0.1, 0.9, n)
        edges = np.linspace(0.1, 0.9, nb + 1) if nb > 1 else None
        return oracle.sort_catalog(phi, np.pi / 2 - theta, z, None, patch, P, edges, "right")

    c1, c2 = polar(n1, B), polar(n2, 1)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    lim = oracle.parse_ang_limits([1.0 * np.pi / 10800], [6.0 * np.pi / 10800])
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)), (B, 1))
    exp, _ = oracle.count_jobs(c1, c2, jobs, t)
    evaluated = {}
    ctx.set_option("strip_width_micro", 0)  # culling along the sort axis alone
    ctx.set_option("auto_orient", 0)        # ... the one the catalogues were uploaded with
    try:
        _sort_axis_checks(ctx, c1, c2, jobs, t, exp, evaluated, P, B)
        # left to itself the library sorts along an axis in the plane of the cap, whatever the upload asked for
        ctx.set_option("auto_orient", 1)
        d1 = _lib.DeviceCatalog(ctx, c1["x"], c1["y"], c1["z"], None, P, B, c1["off"], sort_axis=2)
        d2 = _lib.DeviceCatalog(ctx, c2["x"], c2["y"], c2["z"], None, P, 1, c2["off"], sort_axis=2)
        for kernel in ("band", "sweep"):
            counts, _, st = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel)
            assert np.array_equal(counts, exp)
        assert st.evaluated_pairs <= 1.05 * min(evaluated[0], evaluated[1])
    finally:
        ctx.set_option("auto_orient", 1)
        ctx.set_option("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)


def _sort_axis_checks(ctx, c1, c2, jobs, t, exp, evaluated, P, B):
    from yet_another_wizz_amd import _lib

    for axis in (0, 1, 2):
        d1 = _lib.DeviceCatalog(ctx, c1["x"], c1["y"], c1["z"], None, P, B, c1["off"], sort_axis=axis)
        d2 = _lib.DeviceCatalog(ctx, c2["x"], c2["y"], c2["z"], None, P, 1, c2["off"], sort_axis=axis)
        exp_self, _ = oracle.count_jobs(c1, c1, jobs, t)
        for kernel, kid in (("band", _lib.KERNEL_BAND), ("sweep", _lib.KERNEL_SWEEP)):
            counts, _, st = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel)
            assert np.array_equal(counts, exp), (axis, kernel)
            assert st.kernel_used == kid
            evaluated[axis] = st.evaluated_pairs
            counts, _, _ = _lib.count_pairs(ctx, d1, d1, jobs, t, kernel=kernel)  # self count, non-merged path
            assert np.array_equal(counts, exp_self), (axis, kernel)
    assert evaluated[2] > 2 * min(evaluated[0], evaluated[1])  # z culls worst on a polar cap
    dz = _lib.DeviceCatalog(ctx, c1["x"], c1["y"], c1["z"], None, P, B, c1["off"], sort_axis=2)
    dx = _lib.DeviceCatalog(ctx, c2["x"], c2["y"], c2["z"], None, P, 1, c2["off"], sort_axis=0)
    for kernel in ("sweep", "band"):
        counts, _, st = _lib.count_pairs(ctx, dz, dx, jobs, t, kernel=kernel)
        assert np.array_equal(counts, exp) and st.kernel_used == _lib.KERNEL_FILTER and st.evaluated_pairs == st.candidate_pairs
    with pytest.raises(_lib.YawhipError, match="sort_axis"):
        _lib.DeviceCatalog(ctx, c2["x"], c2["y"], c2["z"], None, P, 1, c2["off"], sort_axis=3)


@pytest.mark.parametrize("kernel", KERNELS)
def test_wide_angles_and_ragged_tiles(ctx, kernel):
    """Separations beyond 90 degrees (the dot-product threshold of the pre-filter becomes negative) and
    segment sizes that are no multiple of the tile: padded lanes must stay silent."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(123)
    def sphere(n, nb):
        ra = rng.uniform(0, 2 * np.pi, n); dec = np.arcsin(rng.uniform(-1, 1, n))
        patch = rng.integers(0, 2, n); z = rng.uniform(0.1, 0.9, n)
        return oracle.sort_catalog(ra, dec, z, None, patch, 2, np.linspace(0.1, 0.9, nb + 1) if nb > 1 else None, "right")
    c1, c2 = sphere(777, 3), sphere(1301, 1)
    jobs = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=np.int32)
    lim = oracle.parse_ang_limits(np.deg2rad([5.0, 60.0]), np.deg2rad([60.0, 150.0]))
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)), (3, 1))
    exp, _ = oracle.count_jobs(c1, c2, jobs, t)
    counts, _, _ = _lib.count_pairs(ctx, _upload(ctx, c1), _upload(ctx, c2), jobs, t, kernel=kernel)
    assert np.array_equal(counts, exp) and exp.sum() > 0.5 * 777 * 1301 * 0.3
    # and the self count of the binned catalogue (ordinary items)
    exp, _ = oracle.count_jobs(c1, c1, jobs, t)
    counts, _, _ = _lib.count_pairs(ctx, _upload(ctx, c1), _upload(ctx, c1), jobs, t, kernel=kernel)
    assert np.array_equal(counts, exp)


@pytest.mark.parametrize("weights", ["uu", "ww"])
def test_strip_widths(ctx, weights):
    """The cross-correlation path cuts patches into strips of a global grid and pairs only strips
    within reach. Any grid spacing (also none, and different spacings on the two sides, which
    disables the strip pairing) gives the same fine-bin counts. (That strips reduce the evaluated
    pairs at survey sizes is checked in test_gpu_scale_properties.py.)"""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(1234)
    P, B = 5, 3
    c1 = _random_catalog(rng, 30000, P, B, weights[0] == "w", dense_box=6.0)
    c2 = _random_catalog(rng, 40000, P, 1, weights[1] == "w", dense_box=6.0)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    t = []
    for k in range(B):
        lim = oracle.parse_ang_limits(np.array([1.0]) * np.pi / 10800, np.array([12.0 + 3 * k]) * np.pi / 10800)
        t.append(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)))
    t = np.stack(t)
    exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
    assert exp_c.sum() > 10000
    evaluated = {}
    try:
        for micro in (0, 1000, 3500, 20000, 300000):
            ctx.set_option("strip_width_micro", micro)
            d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
            for tile_r in (0, 1, 4):
                ctx.set_option("tile_r", tile_r)
                for kernel, kid in (("band", _lib.KERNEL_BAND), ("sweep", _lib.KERNEL_SWEEP)):
                    counts, sums, st = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
                    assert st.kernel_used == kid
                    assert np.array_equal(counts, exp_c), (micro, tile_r, kernel)
                    if weights == "ww":
                        np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
            evaluated[micro] = st.evaluated_pairs
            # the sub-slot tables of a call are reused by the next one with the same job list
            for sel in (slice(None), slice(0, 7), slice(0, 7), slice(3, None), slice(None)):
                for kernel in ("band", "sweep"):
                    counts, _, _ = _lib.count_pairs(ctx, d1, d2, jobs[sel], t, kernel=kernel, want_counts=True)
                    assert np.array_equal(counts, exp_c[sel]), (micro, sel, kernel)
            if micro == 20000:  # catalogue with another grid: ordinary (job, bin) items
                ctx.set_option("strip_width_micro", 5000)
                d2b = _upload(ctx, c2)
                for kernel in ("band", "sweep"):
                    counts, _, _ = _lib.count_pairs(ctx, d1, d2b, jobs, t, kernel=kernel, want_counts=True)
                    assert np.array_equal(counts, exp_c)
        with pytest.raises(_lib.YawhipError, match="strip_width_micro"):
            ctx.set_option("strip_width_micro", 10)
    finally:
        ctx.set_option("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)
        ctx.set_option("tile_r", 0)


@pytest.mark.parametrize("seed", range(40))
def test_randomised_configurations(ctx, seed):
    """Small random set-ups against the oracle: random footprint (cap around a random direction, sometimes
    the whole sphere), patch assignment with tiny and empty patches, duplicated objects (equal sort keys and
    zero separations), random number of bins / edges / scales, strip grid spacing and sort axis, weights on
    either side, cross and self counts."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(1000 + seed)
    P = int(rng.integers(1, 7))
    B = int(rng.integers(1, 6))
    n1, n2 = int(rng.integers(50, 3000)), int(rng.integers(50, 3000))
    cap = rng.choice([0.02, 0.2, 2.0])  # half opening (rad) of the footprint; 2.0 ~ most of the sphere
    centre = rng.normal(size=3); centre /= np.linalg.norm(centre)

    def sample(n, nb, weighted):
        v = centre + np.tan(min(cap, 1.5)) * rng.normal(size=(n, 3)) * 0.5
        v /= np.linalg.norm(v, axis=1)[:, None]
        if n > 20:  # duplicates: identical positions
            v[rng.integers(0, n, n // 10)] = v[rng.integers(0, n, n // 10)]
        ra, dec = np.arctan2(v[:, 1], v[:, 0]) % (2 * np.pi), np.arcsin(np.clip(v[:, 2], -1, 1))
        patch = rng.integers(0, P, n)
        if P > 2:
            patch[patch == 1] = 0          # patch 1 stays empty
            patch[: min(2, n)] = P - 1     # a patch with (at least) two objects
        z = rng.uniform(0.0, 1.0, n)
        w = rng.uniform(0.2, 3.0, n) if weighted else None
        edges = np.linspace(0.05, 0.95, nb + 1) if nb > 1 else None
        return oracle.sort_catalog(ra, dec, z, w, patch, P, edges, "right")

    w1, w2 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    nb2 = B if rng.integers(0, 2) else 1
    c1, c2 = sample(n1, B, w1), sample(n2, nb2, w2)
    n_scales = int(rng.integers(1, 4))
    t = []
    for k in range(B):
        lo = np.sort(rng.uniform(0.02, 0.5, n_scales)) * cap * (1 + 0.1 * k)
        hi = lo * rng.uniform(1.5, 6.0, n_scales)
        lim = oracle.parse_ang_limits(lo, np.minimum(hi, 3.0))
        t.append(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)))
    if len({len(x) for x in t}) != 1:  # keep one edge count per call (the ABI takes a rectangular table)
        t = [t[0]] * B
    t = np.stack(t)
    jobs = np.array([(p, q) for p in range(P) for q in range(P) if rng.random() < 0.8 or p == q], dtype=np.int32)
    micro = int(rng.choice([0, 1000, 4000, 20000, 300000]))
    axis = int(rng.integers(0, 3))
    try:
        ctx.set_option("strip_width_micro", micro)
        ctx.set_option("tile_r", int(rng.choice([0, 1, 2, 4])))
        ctx.set_option("band_cap", int(rng.choice([0, 192, 288])))  # LDS stage capacity of the band kernel
        ctx.set_option("hist_copies_log2", int(rng.choice([-1, 0, 2, 4])))  # copies of its LDS histogram
        ctx.set_option("band_batch_log2", int(rng.choice([-1, 0, 1, 3])))  # consecutive items per workgroup visit
        up = lambda c: _lib.DeviceCatalog(ctx, c["x"], c["y"], c["z"], c["w"], P, c["nb"], c["off"], sort_axis=axis)
        d1, d2 = up(c1), up(c2)
        for a, b, da, db in ((c1, c2, d1, d2), (c1, c1, d1, d1)):
            exp_c, exp_s = oracle.count_jobs(a, b, jobs, t)
            for kernel in ("band", "sweep", "exact"):
                counts, sums, _ = _lib.count_pairs(ctx, da, db, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
                assert np.array_equal(counts, exp_c), (seed, kernel)
                if a["w"] is None and b["w"] is None:
                    assert np.array_equal(sums, exp_c.astype(np.float64))
                else:
                    np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
            assert np.array_equal(_lib.job_work(ctx, da, db, jobs, t, kernel="exact").sum(),
                                  _lib.count_pairs(ctx, da, db, jobs, t, kernel="exact")[2].candidate_pairs)
    finally:
        ctx.set_option("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)
        ctx.set_option("tile_r", 0)
        ctx.set_option("band_cap", 0)
        ctx.set_option("hist_copies_log2", -1)
        ctx.set_option("band_batch_log2", -1)


@pytest.mark.parametrize("weights", ["uu", "ww"])
def test_segment_strip_layout_path(ctx, weights):
    """Binned x binned counts on the per-(patch, bin) strip layouts (built for dense catalogues; forced here
    with ``seg_strips_min_run`` = 1): cross and self counts, several tile sizes and grid spacings, against
    the oracle and against the ordinary per-bin items (``seg_strips`` = 0): identical counts (the saving in
    evaluated pairs only shows for dense catalogues)."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(97)
    P, B = 4, 5
    c1 = _random_catalog(rng, 40000, P, B, weights[0] == "w", dense_box=6.0)
    c2 = _random_catalog(rng, 50000, P, B, weights[1] == "w", dense_box=6.0)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    t = []
    for k in range(B):
        lim = oracle.parse_ang_limits(np.array([1.0, 4.0]) * np.pi / 10800, np.array([4.0, 10.0 + 2 * k]) * np.pi / 10800)
        t.append(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)))
    t = np.stack(t)
    try:
        ctx.set_option("seg_strips_min_run", 1)
        evaluated = {}
        for micro in (3000, 8000):
            ctx.set_option("strip_width_micro", micro)
            d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
            for a, b, da, db in ((c1, c2, d1, d2), (c2, c2, d2, d2)):
                exp_c, exp_s = oracle.count_jobs(a, b, jobs, t)
                assert exp_c.sum() > 10000
                for seg in (1, 0):
                    ctx.set_option("seg_strips", seg)
                    for tile_r in (0, 1, 4):
                        ctx.set_option("tile_r", tile_r)
                        for kernel in ("band", "sweep"):
                            counts, sums, st = _lib.count_pairs(ctx, da, db, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
                            assert np.array_equal(counts, exp_c), (micro, seg, tile_r, kernel)
                            if weights == "ww":
                                np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
                        evaluated[(micro, a is c1, seg, tile_r)] = st.evaluated_pairs
                    work = _lib.job_work(ctx, da, db, jobs, t, kernel="sweep")
                    # (a weighted call that also returns counts runs the kernel twice)
                    assert work.sum() * (2 if weights == "ww" else 1) == evaluated[(micro, a is c1, seg, 4)]
    finally:
        ctx.set_option("seg_strips_min_run", 16)
        ctx.set_option("seg_strips", 1)
        ctx.set_option("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)
        ctx.set_option("tile_r", 0)


def _tangent_partners(rng, a, ang, direction):
    """Points at angular distance ``ang`` from the unit vectors ``a`` along the tangent ``direction`` (projected)."""
    v = direction - (direction * a).sum(1, keepdims=True) * a
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    b = a * np.cos(ang)[:, None] + v * np.sin(ang)[:, None]
    return b / np.linalg.norm(b, axis=1, keepdims=True)


@pytest.mark.parametrize("kernel", ["band", "sweep"])
@pytest.mark.parametrize("layout", ["merged", "segments"])
@pytest.mark.parametrize("micro", [None, 2500, 1000])
def test_borderline_pairs_on_the_strip_paths(ctx, kernel, layout, micro):
    """The production paths -- binned x unbinned on the merged (patch, strip) runs, binned x binned on the
    (patch, bin, strip) runs -- with engineered partners at theta_edge(k) (1 +- delta) of the object's OWN redshift bin
    (per-bin scales), displaced along the sort axis (they sit at the edge u +- r_win of the window and of the per-object
    band), along the strip axis (they sit in the neighbouring strips, up to ``reach`` grid cells away) and across the
    patch boundary. Footprint around (1, 0, 0): the library sorts along z and cuts strips along y there. Bit parity
    with the oracle, for the default grid (one cell > theta_max) and for finer grids (reach 2 and 4)."""
    # theta_max = 8 arcmin * 1.7 -> chord 3.96e-3: grid 5e-3 (default) reach 1, 2.5e-3 reach 2, 1e-3 reach 4
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(97 + (micro or 0))
    B, P, n = 3, 2, 6000
    theta0 = 8.0 * np.pi / 10800
    thetas = theta0 * (1.0 + 0.35 * np.arange(B))                       # theta_max of bin k
    t_rows, edges = [], []
    for k in range(B):
        lim = oracle.parse_ang_limits([0.1 * thetas[k]], [thetas[k]])
        ab = oracle.ang_bins_for(lim, None, None)
        edges.append(ab)
        t_rows.append(oracle.thresholds_for(ab))
    t = np.stack(t_rows)
    # c1 objects p: a small box around ra = 0, dec = 0, redshift bin k
    ra = rng.uniform(-0.01, 0.01, n)
    dec = rng.uniform(-0.01, 0.01, n)
    p = np.column_stack(oracle.to_3d(ra % (2 * np.pi), dec))
    kbin = rng.integers(0, B, n)
    z1 = 0.1 + (kbin + rng.uniform(0.05, 0.95, n)) * (0.8 / B)
    which_edge = rng.integers(0, 2, n)                                     # inner or outer edge of the bin's annulus
    edge = np.array([edges[k][w] for k, w in zip(kbin, which_edge)])
    delta = rng.choice([0.0, 1e-16, -1e-16, 3e-16, -3e-16, 1e-14, -1e-14, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6], n)
    ang = edge * (1.0 + delta)
    north, east = np.array([0.0, 0.0, 1.0]), np.array([0.0, 1.0, 0.0])      # +dec = sort axis z, +ra = strip axis y
    style = rng.integers(0, 4, n)
    direction = np.where((style == 0)[:, None], north, np.where((style == 1)[:, None], -north,
                         np.where((style == 2)[:, None], east, -east))) * np.ones((n, 3))
    direction = direction + (rng.random(n) < 0.3)[:, None] * rng.normal(size=(n, 3)) * 0.3  # some oblique ones
    q = _tangent_partners(rng, p, ang, direction)
    ra2, dec2 = np.arctan2(q[:, 1], q[:, 0]), np.arcsin(np.clip(q[:, 2], -1, 1))
    z2 = z1.copy()                                                          # partner in the same redshift bin (segments)
    patch1 = (dec > 0).astype(int)                                          # patch boundary along dec = 0: pairs straddle it
    patch2 = (dec2 > 0).astype(int)
    zedges = np.linspace(0.1, 0.9, B + 1)
    c1 = oracle.sort_catalog(ra % (2 * np.pi), dec, z1, None, patch1, P, zedges, "right")
    c2 = oracle.sort_catalog(ra2 % (2 * np.pi), dec2, z2, None, patch2, P, zedges if layout == "segments" else None, "right")
    jobs = np.array([(a, b) for a in range(P) for b in range(P)], dtype=np.int32)
    exp, _ = oracle.count_jobs(c1, c2, jobs, t)
    assert exp.sum() > n // 4 and (exp[1] + exp[2]).sum() > 50           # engineered pairs dominate; many cross the patch boundary
    try:
        if micro is not None:
            ctx.set_option("strip_width_micro", micro)
        if layout == "segments":
            ctx.set_option("seg_strips_min_run", 1)
        for axis in (2, 0):                                               # the catalogues' own sort axis does not matter
            d1 = _lib.DeviceCatalog(ctx, c1["x"], c1["y"], c1["z"], None, P, B, c1["off"], sort_axis=axis)
            d2 = _lib.DeviceCatalog(ctx, c2["x"], c2["y"], c2["z"], None, P, c2["nb"], c2["off"], sort_axis=axis)
            for tile_r in (0, 1):
                ctx.set_option("tile_r", tile_r)
                counts, _, st = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel)
                assert st.layout_mode == (1 if layout == "merged" else 3)
                assert np.array_equal(counts, exp), (axis, tile_r)
            # swapped roles (unbinned x binned: plain layout) count the same pairs
            counts, _, _ = _lib.count_pairs(ctx, d2, d1, jobs[:, ::-1].copy(), t, kernel=kernel)
            assert np.array_equal(counts, exp)
    finally:
        ctx.set_option("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)
        ctx.set_option("seg_strips_min_run", 16)
        ctx.set_option("tile_r", 0)


@pytest.mark.parametrize("theta_deg", [8.0 / 60.0, 3.0, 30.0])
@pytest.mark.parametrize("grid", ["annulus", "two_scales", "fine"])
def test_float32_guard_bands_are_decided_exactly(ctx, theta_deg, grid):
    """The band kernel classifies in float32 and hands evaluations inside a guard band |s32 - t| <= g(t) to the exact
    float64 predicate. Partners are engineered at theta_edge (1 +- delta) with delta log-uniform in [1e-10, 1e-3] (and
    exactly on the edge): they fall inside the guard bands, at their borders and just outside, for one annulus, for
    overlapping scales (4 edges) and for a fine log-spaced grid (25 edges: k_count_band32_fine), from arcminutes (the
    sqrt(t) term of g dominates) to tens of degrees (the relative term). Bit parity with the oracle."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(int(theta_deg * 1000) + len(grid))
    B, P, n = 3, 2, 5000
    theta0 = np.deg2rad(theta_deg)
    t_rows, edges = [], []
    for k in range(B):
        th = theta0 * (1.0 + 0.35 * k)
        if grid == "annulus":
            ab = oracle.ang_bins_for(oracle.parse_ang_limits([0.1 * th], [th]), None, None)
        elif grid == "two_scales":
            ab = oracle.ang_bins_for(oracle.parse_ang_limits([0.1 * th, 0.3 * th], [0.4 * th, th]), None, None)
        else:
            ab = oracle.ang_bins_for(oracle.parse_ang_limits([0.1 * th], [th]), -1.0, 24)
        edges.append(ab)
        t_rows.append(oracle.thresholds_for(ab))
    t = np.stack(t_rows)
    assert t.shape[1] == {"annulus": 2, "two_scales": 4, "fine": 25}[grid]
    ra = rng.uniform(-0.01, 0.01, n)
    dec = rng.uniform(-0.01, 0.01, n)
    p = np.column_stack(oracle.to_3d(ra % (2 * np.pi), dec))
    kbin = rng.integers(0, B, n)
    z1 = 0.1 + (kbin + rng.uniform(0.05, 0.95, n)) * (0.8 / B)
    edge = np.array([edges[k][rng.integers(0, len(edges[k]))] for k in kbin])
    delta = np.where(rng.random(n) < 0.1, 0.0, rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-10.0, -3.0, n))
    ang = edge * (1.0 + delta)
    q = _tangent_partners(rng, p, ang, rng.normal(size=(n, 3)))
    ra2, dec2 = np.arctan2(q[:, 1], q[:, 0]), np.arcsin(np.clip(q[:, 2], -1, 1))
    zedges = np.linspace(0.1, 0.9, B + 1)
    c1 = oracle.sort_catalog(ra % (2 * np.pi), dec, z1, None, (dec > 0).astype(int), P, zedges, "right")
    c2 = oracle.sort_catalog(ra2 % (2 * np.pi), dec2, None, None, (dec2 > 0).astype(int), P, None, "right")
    jobs = np.array([(a, b) for a in range(P) for b in range(P)], dtype=np.int32)
    exp, _ = oracle.count_jobs(c1, c2, jobs, t)
    assert exp.sum() > n // 4
    micro = int(min(max(np.ceil(1.02e6 * np.sqrt(t.max()) / 50.0) * 50.0, 1000), 2000000))
    try:
        ctx.set_option("strip_width_micro", micro)
        d1 = _lib.DeviceCatalog(ctx, c1["x"], c1["y"], c1["z"], None, P, B, c1["off"])
        d2 = _lib.DeviceCatalog(ctx, c2["x"], c2["y"], c2["z"], None, P, 1, c2["off"])
        for fp32 in (1, 0):
            ctx.set_option("band_fp32", fp32)
            counts, _, st = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel="band")
            assert st.kernel_used == _lib.KERNEL_BAND and st.layout_mode == 1
            assert np.array_equal(counts, exp), fp32
            # the engineered partners sit inside the guard bands: the exact predicate must have been called on for many of them
            # (a fine grid over tens of degrees is not exactly log-spaced in chord^2 any more -- sin --, but the model only has
            # to name the nearest edge: it stays on the fine-grid kernel)
            want = 64 if not fp32 else (33 if grid == "fine" else 32)
            assert st.band_variant == want
            assert (st.exact_reevaluations > n // 20) if want != 64 else (st.exact_reevaluations == 0)
    finally:
        ctx.set_option("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)
        ctx.set_option("band_fp32", 1)


@pytest.mark.parametrize("kernel", ["exact", "filter"])
def test_weighted_shared_histogram_corner(ctx, kernel):
    """Weighted brute-force counts with more fine bins than per-lane private float64 histograms fit in LDS
    ((E - 1) * 256 * 8 B > 160 KB for E - 1 > 78): the workgroup-shared float64 histogram (LDS atomics across four
    waves). Sums within 1e-10 of the oracle (that corner is not bit-reproducible, include/yawhip.h says so)."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(404)
    c1 = _random_catalog(rng, 3000, 2, 2, True, dense_box=1.0)
    c2 = _random_catalog(rng, 2500, 2, 1, True, dense_box=1.0)
    jobs = np.array([[0, 0], [0, 1], [1, 1]], dtype=np.int32)
    lim = oracle.parse_ang_limits(np.array([0.5]) * np.pi / 10800, np.array([40.0]) * np.pi / 10800)
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(lim, -1.0, 95)), (2, 1))
    assert t.shape[1] - 1 > 78
    exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
    counts, sums, st = _lib.count_pairs(ctx, _upload(ctx, c1), _upload(ctx, c2), jobs, t, kernel=kernel, want_counts=True, want_sums=True)
    assert np.array_equal(counts, exp_c) and exp_c.sum() > 1e5 and (exp_c > 0).sum() > 0.5 * exp_c.size
    np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
    for k2 in ("band", "sweep"):  # the strip paths hold the same histogram per wave
        counts, sums, _ = _lib.count_pairs(ctx, _upload(ctx, c1), _upload(ctx, c2), jobs, t, kernel=k2, want_counts=True, want_sums=True)
        assert np.array_equal(counts, exp_c)
        np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)


def test_histogram_too_large_for_one_item_falls_back_to_per_bin_items(ctx):
    """400 z-bins x 51 separation-weight bins, weighted: the all-bins-in-one-item histogram (163 KB) does not fit the LDS.
    The count must fall back to (job, bin) items -- same results, ``layout_mode`` 0 -- instead of failing."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(77)
    B = 400
    c1 = _random_catalog(rng, 6000, 3, B, True, dense_box=1.0)
    c2 = _random_catalog(rng, 5000, 3, 1, True, dense_box=1.0)
    jobs = np.array([[0, 0], [0, 1], [1, 2], [2, 2]], dtype=np.int32)
    lim = oracle.parse_ang_limits(np.array([0.5]) * np.pi / 10800, np.array([60.0]) * np.pi / 10800)
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(lim, -1.0, 50)), (B, 1))
    exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
    assert exp_c.sum() > 1e4
    d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
    for kernel in ("auto", "band", "sweep"):
        counts, sums, st = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
        assert np.array_equal(counts, exp_c), kernel
        np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
        assert st.layout_mode == 0
    # with a quarter of the bins the merged item fits again
    t4 = t[: B // 4]
    c1b = _random_catalog(rng, 6000, 3, B // 4, True, dense_box=1.0)
    exp_c, exp_s = oracle.count_jobs(c1b, c2, jobs, t4)
    counts, sums, st = _lib.count_pairs(ctx, _upload(ctx, c1b), d2, jobs, t4, kernel="auto", want_counts=True, want_sums=True)
    assert np.array_equal(counts, exp_c) and st.layout_mode == 1
    np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)


def test_counts_beyond_32_bits_per_item(ctx):
    """More than 2^32 pairs in ONE (work item, fine bin): 4.3 M streamed objects against a tile of 1024 lane objects,
    all inside one wide bin. The 32-bit LDS counters are moved to the 64-bit result before they wrap."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(2)
    n1, n2 = 4_300_000, 1024

    def clump(n):
        ra = np.deg2rad(50.0 + rng.normal(0, 0.02, n)); dec = np.deg2rad(10.0 + rng.normal(0, 0.02, n))
        return oracle.sort_catalog(ra, dec, None, None, np.zeros(n, dtype=int), 1, None, "right")
    c1, c2 = clump(n1), clump(n2)
    t = np.array([[0.0, 3.9]])  # every pair but coincident ones
    d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
    ctx.set_option("tile_r", 4)
    try:
        for kernel in ("exact", "filter", "sweep", "band"):
            counts, _, st = _lib.count_pairs(ctx, d1, d2, [[0, 0]], t, kernel=kernel)
            assert counts[0, 0, 0] == n1 * n2 > 2**32, kernel
    finally:
        ctx.set_option("tile_r", 0)


@pytest.mark.parametrize("weights", ["uu", "ww"])
@pytest.mark.parametrize("grid", ["annulus", "two_scales", "fine"])
def test_windows_longer_than_the_stage(ctx, grid, weights):
    """A dense streamed side: the window of a lane tile holds ~1500 entries, several times the LDS stage of the band kernels
    (320 / 512 entries, 192 / 288 for fine grids): it goes through the stage in pieces, the bands of a lane continue from
    piece to piece, and the counters are flushed per round when ``flush_stages_log2`` says so. Merged items (binned x
    unbinned) and per-bin items (binned x binned); bit parity / 1e-10 with the oracle."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(31 + len(grid))
    B = 3
    def cat(n, nb, w):
        ra = np.deg2rad(rng.uniform(80.0, 80.5, n)); dec = np.deg2rad(rng.uniform(-20.0, -19.5, n))
        z = rng.uniform(0.1, 0.9, n)
        return oracle.sort_catalog(ra, dec, z, rng.uniform(0.5, 1.5, n) if w else None, (ra > np.deg2rad(80.25)).astype(int), 2,
                                   np.linspace(0.1, 0.9, nb + 1) if nb > 1 else None, "right")
    c1 = cat(12000, B, weights[0] == "w")
    th = 3.0 * np.pi / 10800
    if grid == "annulus":
        ab = oracle.ang_bins_for(oracle.parse_ang_limits([0.1 * th], [th]), None, None)
    elif grid == "two_scales":
        ab = oracle.ang_bins_for(oracle.parse_ang_limits([0.1 * th, 0.3 * th], [0.4 * th, th]), None, None)
    else:
        ab = oracle.ang_bins_for(oracle.parse_ang_limits([0.1 * th], [th]), -1.0, 16)
    t = np.tile(oracle.thresholds_for(ab), (B, 1))
    jobs = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=np.int32)
    micro = int(min(max(np.ceil(1.02e6 * np.sqrt(t.max()) / 50.0) * 50.0, 1000), 2000000))
    try:
        ctx.set_option("strip_width_micro", micro)
        ctx.set_option("seg_strips_min_run", 1)
        d1 = _lib.DeviceCatalog(ctx, c1["x"], c1["y"], c1["z"], c1["w"], 2, B, c1["off"])
        for nb2, n2 in ((1, 60000), (B, 90000)):
            c2 = cat(n2, nb2, weights[1] == "w")
            d2 = _lib.DeviceCatalog(ctx, c2["x"], c2["y"], c2["z"], c2["w"], 2, nb2, c2["off"])
            exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
            assert exp_c.sum() > 1e6
            # (objects per lane, merged triple runs): the library's choice; one / two objects per lane with the merged window --
            # three times as long -- forced through the stage in pieces (one object per lane: two entries per trip of the walk)
            for tile_r, triple in ((0, 1), (1, 2), (2, 2)):
                ctx.set_option("tile_r", tile_r)
                ctx.set_option("triple_runs", triple)
                for log2 in (17, 0):
                    ctx.set_option("flush_stages_log2", log2)
                    counts, sums, st = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel="band", want_counts=True, want_sums=True)
                    assert st.kernel_used == _lib.KERNEL_BAND and st.layout_mode == (1 if nb2 == 1 else 3)
                    assert st.band_variant == (33 if grid == "fine" else 32)
                    assert np.array_equal(counts, exp_c), (nb2, log2, tile_r, triple)
                    if weights == "uu":
                        assert np.array_equal(sums, exp_c.astype(np.float64))
                    else:
                        np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
    finally:
        ctx.set_option("strip_width_micro", _lib.DEFAULT_STRIP_MICRO)
        ctx.set_option("seg_strips_min_run", 16)
        ctx.set_option("flush_stages_log2", 17)
        ctx.set_option("tile_r", 0)
        ctx.set_option("triple_runs", 1)


def test_band_kernel_flush_interval(ctx):
    """``flush_stages_log2`` = 0 makes the band kernel move its 32-bit LDS counters to the 64-bit result after every
    stage of a long window (the path that otherwise only runs after 2^17 stages): same counts."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(8)
    def clump(n, nb):
        ra = np.deg2rad(200.0 + rng.normal(0, 0.01, n)); dec = np.deg2rad(-45.0 + rng.normal(0, 0.01, n))
        z = rng.uniform(0.1, 0.9, n)
        return oracle.sort_catalog(ra, dec, z, None, np.zeros(n, dtype=int), 1, np.linspace(0.1, 0.9, nb + 1) if nb > 1 else None, "right")
    c1, c2 = clump(3000, 3), clump(700, 1)
    lim = oracle.parse_ang_limits([0.2 * np.pi / 10800], [3.0 * np.pi / 10800])
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)), (3, 1))
    exp, _ = oracle.count_jobs(c1, c2, [[0, 0]], t)
    d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
    try:
        for log2 in (0, 1, 17):
            ctx.set_option("flush_stages_log2", log2)
            counts, _, _ = _lib.count_pairs(ctx, d1, d2, [[0, 0]], t, kernel="band")
            assert np.array_equal(counts, exp), log2
            counts, _, _ = _lib.count_pairs(ctx, d1, d1, [[0, 0]], t, kernel="band")  # per-bin items
            assert np.array_equal(counts, oracle.count_jobs(c1, c1, [[0, 0]], t)[0]), log2
    finally:
        ctx.set_option("flush_stages_log2", 17)
    assert exp.sum() > 0.3 * 3000 * 700


@pytest.mark.parametrize("kernel", ["band", "sweep", "filter"])
def test_weighted_slab_budget_splits_the_job_list(ctx, kernel):
    """A weighted call keeps a slab of partial sums per potential work item. With a small ``slab_budget_bytes`` the
    library counts the job list in pieces; the sums do not change by a bit (each job's items are reduced in the same
    order) and the statistics add up."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(55)
    P, B = 6, 4
    c1 = _random_catalog(rng, 20000, P, B, True, dense_box=3.0)
    c2 = _random_catalog(rng, 25000, P, 1, True, dense_box=3.0)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    lim = oracle.parse_ang_limits(np.array([0.5, 1.58, 5.0]) * np.pi / 10800, np.array([1.58, 5.0, 15.8]) * np.pi / 10800)
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(lim, None, None)), (B, 1))
    d1, d2 = _upload(ctx, c1), _upload(ctx, c2)
    counts0, sums0, st0 = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
    exp_c, exp_s = oracle.count_jobs(c1, c2, jobs, t)
    assert np.array_equal(counts0, exp_c) and exp_c.sum() > 1e5
    np.testing.assert_allclose(sums0, exp_s, rtol=RTOL_W, atol=0)
    try:
        ctx.set_option("slab_budget_bytes", 4096)  # far below one job's slabs: cut down to single jobs
        counts1, sums1, st1 = _lib.count_pairs(ctx, d1, d2, jobs, t, kernel=kernel, want_counts=True, want_sums=True)
    finally:
        ctx.set_option("slab_budget_bytes", 1 << 30)
    assert np.array_equal(counts1, counts0) and np.array_equal(sums1, sums0)
    assert st1.candidate_pairs == st0.candidate_pairs and st1.evaluated_pairs == st0.evaluated_pairs
    assert st1.n_launches > st0.n_launches


@pytest.mark.parametrize("weights", ["uu", "ww"])
@pytest.mark.parametrize("lattice,n", [(False, 60000), (True, 60000), (False, 300000)])
def test_self_counts_take_every_pair_once_on_the_diagonal(ctx, weights, lattice, n):
    """Half bands (round 4): a catalogue counted against itself on merged triple runs with one object per lane meets every
    unordered pair of a diagonal job from ONE side -- the lane walks only the entries behind its own place in the triple run of
    its strip -- and counts it twice. Against the oracle's ordered-pair counts (src/yaw/catalog/trees.py:303-362 with tree ==
    other), with the option off, and on a LATTICE: thousands of objects share their sort key exactly (the order inside the
    triples must be one total order of objects: key, run, place in the run), many lie exactly on strip boundaries and every
    object exists twice (pairs at distance zero belong to no bin)."""
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(4242)
    P, B = 3, 4
    # (300 000 objects: merged windows of ~400 entries go through the 320-entry stage in PIECES -- a lane's own place then lies
    # in front of, inside or behind the piece in hand)
    if lattice:
        g = int(np.sqrt(n / 2))
        ra, dec = np.meshgrid(np.deg2rad(50.0 + 5.0 * np.arange(g) / g), np.deg2rad(-2.5 + 5.0 * np.arange(g) / g))
        ra, dec = np.tile(ra.ravel(), 2), np.tile(dec.ravel(), 2)  # every object twice
    else:
        ra = np.deg2rad(rng.uniform(50.0, 55.0, n))
        dec = np.arcsin(rng.uniform(np.sin(np.deg2rad(-2.5)), np.sin(np.deg2rad(2.5)), n))
    m = len(ra)
    patch = np.minimum((np.rad2deg(ra) - 50.0) / 5.0 * P, P - 1).astype(np.int64)
    z = rng.uniform(0.1, 0.9, m)
    w = rng.uniform(0.5, 1.5, m) if weights == "ww" else None
    cat = oracle.sort_catalog(ra, dec, z, w, patch, P, np.linspace(0.1, 0.9, B + 1), "right")
    jobs = np.array([(p, q) for p in range(P) for q in range(p, P)], dtype=np.int32)  # i <= j, as an autocorrelation lists them
    lim = oracle.parse_ang_limits(np.array([1.0]) * np.pi / 10800, np.array([9.0]) * np.pi / 10800)
    t = np.stack([oracle.thresholds_for(oracle.ang_bins_for(lim, None, None))] * B)
    exp_c, exp_s = oracle.count_jobs(cat, cat, jobs, t)
    diag = jobs[:, 0] == jobs[:, 1]
    assert exp_c[diag].sum() > 50000 and np.all(exp_c[diag] % 2 == 0)
    try:
        ctx.set_option("seg_strips_min_run", 1)
        ctx.set_option("strip_width_micro", 3000)
        ctx.set_option("triple_runs", 2)  # merged triple runs whatever the window estimate says
        dev = _upload(ctx, cat)
        seen = {}
        for half in (1, 0):
            ctx.set_option("half_bands", half)
            for tile_r in (1, 2, 0):
                ctx.set_option("tile_r", tile_r)
                counts, sums, st = _lib.count_pairs(ctx, dev, dev, jobs, t, kernel="band", want_counts=True, want_sums=True)
                assert st.band_variant == 32 and st.merged_triples == 1 and st.layout_mode == 3
                assert np.array_equal(counts, exp_c), (half, tile_r)
                if weights == "ww":
                    np.testing.assert_allclose(sums, exp_s, rtol=RTOL_W, atol=0)
                seen[(half, tile_r)] = st.evaluated_pairs
        # one object per lane: the diagonal jobs (most of the pairs) walk half their bands
        assert seen[(1, 1)] < 0.7 * seen[(0, 1)]
        assert seen[(1, 2)] == seen[(0, 2)]  # two objects per lane: unchanged
    finally:
        for key, value in (("seg_strips_min_run", 16), ("strip_width_micro", _lib.DEFAULT_STRIP_MICRO), ("triple_runs", 1),
                           ("half_bands", 1), ("tile_r", 0)):
            ctx.set_option(key, value)
