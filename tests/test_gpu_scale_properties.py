"""GPU: BASELINE.json-sized inputs checked through size-independent properties (the oracle would
take hours here): equality of the three device code paths, role-swap symmetry, additivity over a
split catalogue, invariance under the patch decomposition, exact weight scaling, ordered-pair
symmetry of self counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P, B = 16, 30
ARCMIN = np.pi / 10800


def _box_catalog(seed, n, with_z, width=60.0, height=30.0, weights=None, draw_weights=False):
    """BASELINE config #2's footprint: a 60 x 30 degree box, 16 patches on a regular 4 x 4 grid --
    bench.box_sky / bench.box_centers, the recipe tools/time_reference.py --box fed to the reference."""
    import bench
    import yet_another_wizz_amd as yaw

    ra, dec, rng = bench.box_sky(seed, n, width, height)
    centers = yaw.AngularCoordinates(bench.box_centers(width, height, 4))
    z = rng.uniform(0.1, 1.0, n) if with_z else None
    if draw_weights:  # as tools/time_reference.py --weights: drawn after the redshifts
        weights = rng.uniform(0.5, 1.5, n)
    # radian in, as the reference's run (deg2rad of the recipe's degrees: the same doubles on both sides)
    return yaw.Catalog.from_arrays(np.deg2rad(ra), np.deg2rad(dec), redshifts=z, weights=weights, patch_centers=centers,
                                   degrees=False), (ra, dec, z)


@pytest.fixture(scope="module")
def setup():
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine
    from yet_another_wizz_amd.measurements import angular_plans, threshold_table

    n = 1_000_000
    ref, ref_cols = _box_catalog(101, n, True)
    unk, unk_cols = _box_catalog(202, n, False)
    config = yaw.Configuration.create(rmin=1.0, rmax=10.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=B)
    lref = ref.build_trees(config.binning.edges, closed=config.binning.closed)
    lunk = unk.build_trees(None)
    t = threshold_table(angular_plans(config))
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    fine, stats = engine.count_fine(lref, lunk, jobs, t, kernel="sweep")
    return dict(config=config, ref=ref, unk=unk, lref=lref, lunk=lunk, t=t, jobs=jobs, fine=fine, stats=stats,
                ref_cols=ref_cols, unk_cols=unk_cols)


@pytest.mark.parametrize("weighted", [False, True])
def test_config2_every_slot_against_the_reference(setup, weighted):
    """BASELINE config #2 at its size against the reference ITSELF: ``tests/golden/fullsize_slots_config2*.npz`` hold the
    values ``counts[:, i, j]`` of every linked patch pair and the ``sum_weights`` that the reference's
    ``PatchLinkage.count_pairs`` produced for these inputs (tools/time_reference.py --n-ref 1e6 --n-unk 1e6 --patches 16
    --box 60x30 [--weights]; src/yaw/correlation/measurements.py:354-364). The AUTO path through the public entry point:
    exact (unweighted) / 1e-10 (weighted), nothing outside the linked pairs."""
    import os

    import yet_another_wizz_amd as yaw
    from conftest import GOLDEN
    from yet_another_wizz_amd import _lib

    slots = np.load(os.path.join(GOLDEN, "fullsize_slots_config2" + ("_weighted" if weighted else "") + ".npz"))
    assert int(slots["n_ref"]) == 1_000_000 and int(slots["patches"]) == P and tuple(slots["box"]) == (60.0, 30.0)
    if weighted:
        ref, _ = _box_catalog(101, 1_000_000, True, draw_weights=True)
        unk, _ = _box_catalog(202, 1_000_000, False, draw_weights=True)
        ref.build_trees(setup["config"].binning.edges, closed=setup["config"].binning.closed)
        unk.build_trees(None)
    else:
        ref, unk = setup["ref"], setup["unk"]
    links = yaw.PatchLinkage.from_catalogs(setup["config"], ref, unk)
    jobs = links.get_patch_pairs(ref, unk)
    ids, values = slots["DD_ids"], slots["DD_values"]  # [n, 2], [n, 1, B]
    assert {tuple(j) for j in jobs.tolist()} == {tuple(j) for j in ids.tolist()}  # the reference's linkage
    (res,) = links.count_pairs(ref, unk)
    st = links.last_stats
    assert st.kernel_used == _lib.KERNEL_BAND and st.band_variant == 32  # the default path, not a chosen kernel
    assert float(slots["candidate_pairs"]) == float(st.candidate_pairs)
    got = res.counts.counts[:, ids[:, 0], ids[:, 1]].T  # [n, B]
    if weighted:
        np.testing.assert_allclose(got, values[:, 0], rtol=1e-10, atol=0)
        np.testing.assert_allclose(res.sum_weights.sum_weights1, slots["DD_sum_weights1"], rtol=1e-12, atol=0)
        np.testing.assert_allclose(res.sum_weights.sum_weights2, slots["DD_sum_weights2"], rtol=1e-12, atol=0)
    else:
        assert np.array_equal(got, values[:, 0])
        assert np.array_equal(res.sum_weights.sum_weights1, slots["DD_sum_weights1"])
        assert np.array_equal(res.sum_weights.sum_weights2, slots["DD_sum_weights2"])
        # and the module's all-pairs job table (unlinked pairs included) on the sweep kernel holds the same numbers
        full = setup["fine"].reshape(P, P, B)
        assert np.array_equal(full[ids[:, 0], ids[:, 1]], values[:, 0])
    assert values.sum() > 4e7
    mask = np.ones((P, P), dtype=bool)
    mask[ids[:, 0], ids[:, 1]] = False
    assert not res.counts.counts[:, mask].any()  # nothing outside the linked pairs
    if weighted:
        ref.drop_layouts()
        unk.drop_layouts()


def test_three_code_paths_agree_at_1m(setup):
    from yet_another_wizz_amd import engine

    s = setup
    assert s["stats"].candidate_pairs > 9e11 and s["stats"].evaluated_pairs < 0.05 * s["stats"].candidate_pairs
    assert s["fine"].sum() > 4e7
    f_filter, st = engine.count_fine(s["lref"], s["lunk"], s["jobs"], s["t"], kernel="filter")
    assert st.evaluated_pairs == st.candidate_pairs
    assert np.array_equal(f_filter, s["fine"])
    f_band, st = engine.count_fine(s["lref"], s["lunk"], s["jobs"], s["t"], kernel="band")
    assert np.array_equal(f_band, s["fine"])
    # the band kernel evaluates a few times the true pairs, far fewer than the window rectangles of the sweep
    assert f_band.sum() < st.evaluated_pairs < 4 * f_band.sum() and st.evaluated_pairs < 0.5 * s["stats"].evaluated_pairs
    sub = s["jobs"][::5]  # plain FP64 brute force on every 5th job
    f_exact, _ = engine.count_fine(s["lref"], s["lunk"], sub, s["t"], kernel="exact")
    assert np.array_equal(f_exact, s["fine"][::5])


def test_strips_cull_but_do_not_change_counts(setup):
    """Fresh uploads without the strip grid and with other spacings: same counts, fewer evaluated pairs."""
    from yet_another_wizz_amd import _lib, engine

    s = setup
    ctx = engine.get_context()
    evaluated = {"auto": s["stats"].evaluated_pairs}
    for micro in (0, 2000, 20000):
        devs = [_lib.DeviceCatalog(ctx, l.x, l.y, l.z, l.w, l.num_patches, l.num_bins, l.offsets, sort_axis=2,
                                   strip_micro=micro) for l in (s["lref"], s["lunk"])]
        counts, _, st = _lib.count_pairs(ctx, devs[0], devs[1], s["jobs"], s["t"], kernel="sweep")
        assert np.array_equal(counts.astype(np.float64), s["fine"]), micro
        evaluated[micro] = st.evaluated_pairs
        for d in devs:
            d.free()
    assert engine.strip_micro_for(s["t"]) == 3000  # just above the chord of 10 arcmin (2909)
    assert max(evaluated["auto"], evaluated[2000], evaluated[20000]) < 0.6 * evaluated[0]
    assert evaluated["auto"] < evaluated[20000]


def test_job_work_is_what_the_device_evaluates(setup):
    """``yawhip_job_work`` (the cost the host balances over GPUs): per-job figures add up to the
    evaluated pairs of the real call, equal N1*N2 per job for the brute-force kernel, and an LPT
    partition over them is a partition."""
    from yet_another_wizz_amd import engine, parallel

    s = setup
    work = engine.job_work(s["lref"], s["lunk"], s["jobs"], s["t"], kernel="sweep")
    assert work.dtype == np.int64 and work.shape == (len(s["jobs"]),)
    assert work.sum() == s["stats"].evaluated_pairs and work.min() >= 0
    diag = s["jobs"][:, 0] == s["jobs"][:, 1]
    assert work[diag].mean() > 5 * work[~diag].mean()  # neighbours only share a boundary
    brute = engine.job_work(s["lref"], s["lunk"], s["jobs"], s["t"], kernel="exact")
    n1 = s["lref"].segment_sizes()[s["jobs"][:, 0]].sum(axis=1)
    n2 = s["lunk"].segment_sizes()[s["jobs"][:, 1]].sum(axis=1)
    assert np.array_equal(brute, n1 * n2) and brute.sum() == s["stats"].candidate_pairs
    parts = parallel.partition_jobs(work.astype(float), 8)
    assert sorted(np.concatenate(parts).tolist()) == list(range(len(work)))
    loads = np.array([work[p].sum() for p in parts], dtype=float)
    assert loads.max() / loads.mean() < 1.15
    again = engine.job_work(s["lref"], s["lunk"], s["jobs"], s["t"], kernel="sweep")
    assert np.array_equal(work, again)  # exact function of the inputs: every rank derives the same partition


def test_role_swap_symmetry(setup):
    """count(ref_p, unk_q) == count(unk_q, ref_p): lanes <-> stream, binned <-> unbinned."""
    from yet_another_wizz_amd import engine

    s = setup
    swapped, _ = engine.count_fine(s["lunk"], s["lref"], s["jobs"][:, ::-1].copy(), s["t"], kernel="sweep")
    assert np.array_equal(swapped, s["fine"])


def test_additivity_over_split_catalogue(setup):
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine

    s = setup
    ra, dec, _ = s["unk_cols"]
    half = len(ra) // 3
    centers = s["unk"].get_centers()
    total = np.zeros_like(s["fine"])
    for sl in (slice(0, half), slice(half, None)):
        part = yaw.Catalog.from_arrays(ra[sl], dec[sl], patch_centers=s["ref"])
        assert part.num_patches == P and centers is not None
        f, _ = engine.count_fine(s["lref"], part.build_trees(None), s["jobs"], s["t"])
        total += f
        part.drop_layouts()
    assert np.array_equal(total, s["fine"])


def test_invariance_under_patch_decomposition(setup):
    """Summing all patch pairs gives the count of the undivided catalogues."""
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine

    s = setup
    one = yaw.AngularCoordinates(np.deg2rad([[15.0, 15.0]]))
    ra, dec, z = s["ref_cols"]
    ref1 = yaw.Catalog.from_arrays(ra, dec, redshifts=z, patch_centers=one)
    ra, dec, _ = s["unk_cols"]
    unk1 = yaw.Catalog.from_arrays(ra, dec, patch_centers=one)
    f, st = engine.count_fine(ref1.build_trees(s["config"].binning.edges), unk1.build_trees(None), [[0, 0]], s["t"])
    assert np.array_equal(f[0], s["fine"].sum(axis=0))
    assert st.candidate_pairs == len(ra) * ref1.build_trees(s["config"].binning.edges).num_records
    ref1.drop_layouts()
    unk1.drop_layouts()


def test_constant_weights_scale_exactly(setup):
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine

    s = setup
    ra, dec, _ = s["unk_cols"]
    wunk = yaw.Catalog.from_arrays(ra, dec, weights=np.full(len(ra), 2.0), patch_centers=s["ref"])
    f, _ = engine.count_fine(s["lref"], wunk.build_trees(None), s["jobs"], s["t"])
    assert np.array_equal(f, 2.0 * s["fine"])  # powers of two: exact in float64 whatever the summation order
    f2, _ = engine.count_fine(s["lref"], wunk.build_trees(None), s["jobs"], s["t"])
    assert np.array_equal(f, f2)  # run-to-run reproducible
    wunk.drop_layouts()


def test_self_count_is_symmetric_and_even(setup):
    """Auto count of the binned reference: ordered pairs -> counts[p,q] == counts[q,p], diagonal even."""
    from yet_another_wizz_amd import engine

    s = setup
    f, st = engine.count_fine(s["lref"], s["lref"], s["jobs"], s["t"])
    full = np.zeros((P, P) + f.shape[1:])
    full[s["jobs"][:, 0], s["jobs"][:, 1]] = f
    assert np.array_equal(full, full.transpose(1, 0, 2, 3))
    diag = full[np.arange(P), np.arange(P)]
    assert np.all(diag % 2 == 0) and diag.sum() > 1e5
    f_exact, _ = engine.count_fine(s["lref"], s["lref"], s["jobs"], s["t"], kernel="exact")
    assert np.array_equal(f, f_exact)


def test_headline_size_paths_agree():
    """10M x 10M / 64 patches (BASELINE.json configs[2]): sweep == filter on a sample of jobs."""
    import types

    import bench
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine
    from yet_another_wizz_amd.measurements import angular_plans, threshold_table

    args = types.SimpleNamespace(n_ref=10e6, n_unk=10e6, patches=64, zbins=30)
    config, ref, unk = bench.make_catalogs(args)
    lref = ref.build_trees(config.binning.edges, closed=config.binning.closed)
    lunk = unk.build_trees(None)
    links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
    jobs = links.get_patch_pairs(ref, unk)
    t = threshold_table(angular_plans(config))
    f_sweep, st = engine.count_fine(lref, lunk, jobs, t, kernel="sweep")
    assert st.candidate_pairs > 1e13 and f_sweep.sum() > 1e8
    sample = jobs[::40]
    f_filter, _ = engine.count_fine(lref, lunk, sample, t, kernel="filter")
    assert np.array_equal(f_filter, f_sweep[::40])
    # every object pair closer than theta_max is covered by the linkage: unlinked patch pairs count zero
    unlinked = np.array([(p, q) for p in range(0, 64, 9) for q in range(64) if q not in links.patch_links[p]][:40], dtype=np.int32)
    f_un, _ = engine.count_fine(lref, lunk, unlinked, t, kernel="sweep")
    assert f_un.sum() == 0
    ref.drop_layouts()
    unk.drop_layouts()


def test_headline_size_weighted_paths_agree():
    """The same with per-object weights (w ~ U(0.5, 1.5)) on the DEFAULT path (the band kernel): every slot of the
    reference's own weighted result (``tools/time_reference.py --weights``: 440 linked pairs x 30 bins, 1e-10 relative),
    the FP32-filter path on a sample of jobs (another summation order: 1e-12 relative), run-to-run bit
    reproducibility, and weighted sum / count = <w1 w2> ~ 1."""
    import os
    import types

    import bench
    import yet_another_wizz_amd as yaw
    from conftest import GOLDEN
    from yet_another_wizz_amd import _lib, engine
    from yet_another_wizz_amd.measurements import angular_plans, threshold_table

    args = types.SimpleNamespace(n_ref=10e6, n_unk=10e6, patches=64, zbins=30, weights=True)
    config, ref, unk = bench.make_catalogs(args)
    lref = ref.build_trees(config.binning.edges, closed=config.binning.closed)
    lunk = unk.build_trees(None)
    links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
    jobs = links.get_patch_pairs(ref, unk)
    t = threshold_table(angular_plans(config))
    s_band, st = engine.count_fine(lref, lunk, jobs, t)
    assert st.kernel_used == _lib.KERNEL_BAND and st.layout_mode == 1
    s_again, _ = engine.count_fine(lref, lunk, jobs, t)
    assert np.array_equal(s_band, s_again)
    # the reference, slot by slot
    slots = np.load(os.path.join(GOLDEN, "fullsize_slots_config3_weighted.npz"))
    ids, values = slots["DD_ids"], slots["DD_values"]
    (res,) = links.count_pairs(ref, unk)
    np.testing.assert_allclose(res.counts.counts[:, ids[:, 0], ids[:, 1]].T, values[:, 0], rtol=1e-10, atol=0)
    np.testing.assert_allclose(res.sum_weights.sum_weights1, slots["DD_sum_weights1"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(res.sum_weights.sum_weights2, slots["DD_sum_weights2"], rtol=1e-12, atol=0)
    sample = jobs[::40]
    s_filter, _ = engine.count_fine(lref, lunk, sample, t, kernel="filter")
    np.testing.assert_allclose(s_band[::40], s_filter, rtol=1e-12, atol=0)
    s_sweep, _ = engine.count_fine(lref, lunk, sample, t, kernel="sweep")
    np.testing.assert_allclose(s_band[::40], s_sweep, rtol=1e-12, atol=0)
    ctx = engine.get_context()
    counts, sums, _ = _lib.count_pairs(ctx, engine.device_catalog(lref, ctx), engine.device_catalog(lunk, ctx), sample, t,
                                       want_counts=True, want_sums=True)
    assert np.array_equal(sums, s_band[::40]) and counts.sum() > 2e6
    assert abs(sums.sum() / counts.sum() - 1.0) < 0.01
    ref.drop_layouts()
    unk.drop_layouts()
