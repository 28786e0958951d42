"""GPU: the full drop-in API (Catalog -> crosscorrelate / autocorrelate -> CorrFunc.sample) on the
HIP path against outputs captured from the reference, plus size-independent properties at scale."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["u", "w"])
@pytest.mark.parametrize("cfg,closed", [("s2", "right"), ("s2", "left"), ("rw", "right")])
def test_full_driver_vs_reference_on_gpu(tag, cfg, closed):
    helpers.run_full_case(tag, cfg, closed)


@pytest.mark.parametrize("unit,rmin,rmax", [("kpc/h", [150.0, 600.0], [900.0, 2400.0]), ("Mpc/h", [0.2], [1.5]), ("Mpc", [0.15], [1.2])])
def test_comoving_and_mpc_units_device_vs_oracle(unit, rmin, rmax, monkeypatch):
    """Scales in comoving units (reference src/yaw/cosmology.py:262-285: scale / D_C(z) per redshift bin) and in Mpc through
    the public API on the HIP path, against the CPU oracle fed the same thresholds; the distances themselves are pinned by
    tests/test_cosmology.py."""
    import yet_another_wizz_amd as yaw

    rng = np.random.default_rng(len(unit))
    def frame(n, with_z):
        d = {"ra": rng.uniform(20.0, 26.0, n), "dec": rng.uniform(-3.0, 3.0, n)}
        if with_z:
            d["z"] = rng.uniform(0.15, 0.95, n)
        return d
    ref = yaw.Catalog.from_dataframe(None, frame(7000, True), ra_name="ra", dec_name="dec", redshift_name="z", patch_num=5)
    make = lambda n, z: yaw.Catalog.from_dataframe(None, frame(n, z), ra_name="ra", dec_name="dec", redshift_name="z" if z else None,
                                                   patch_centers=ref)
    unk, rand = make(9000, False), make(14000, False)
    config = yaw.Configuration.create(rmin=rmin, rmax=rmax, unit=unit, zmin=0.2, zmax=0.9, num_bins=5)
    dev = yaw.crosscorrelate(config, ref, unk, unk_rand=rand)
    helpers.use_oracle_engine(monkeypatch)
    ora = yaw.crosscorrelate(config, ref, unk, unk_rand=rand)
    assert len(dev) == len(ora) == len(rmin)
    for cd, co in zip(dev, ora):
        for kind in ("dd", "dr"):
            assert np.array_equal(getattr(cd, kind).counts.counts, getattr(co, kind).counts.counts), (unit, kind)
        assert co.dd.counts.counts.sum() > 500
        np.testing.assert_allclose(cd.sample().data, co.sample().data, rtol=1e-12, atol=1e-14, equal_nan=True)


def test_twodflens_vs_reference_on_gpu():
    helpers.run_twodflens_case()


def test_reference_end_to_end_known_answer_on_gpu():
    helpers.run_reference_example_case()


def test_reference_cache_on_gpu():
    """A cache the reference wrote (data.bin / meta.yml / patch_ids.bin) -> Catalog(path) -> crosscorrelate and
    autocorrelate on the HIP path, against the reference's own measurements from that cache."""
    from yet_another_wizz_amd import engine

    helpers.run_refcache_case()
    assert engine.get_context() is not None


def test_native_library_is_the_one_running():
    from yet_another_wizz_amd import _lib

    assert _lib.device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libyawhip.so" in f.read()


@pytest.mark.parametrize("seed", range(6))
def test_random_measurements_device_vs_oracle(seed, monkeypatch):
    """The whole public path on random inputs: crosscorrelate (DD, DR, RD, RR) and autocorrelate (DD, DR, RR)
    with random footprints, patch counts (k-means and given centres), scales in angular and physical units,
    rweight on/off, closed left/right and weights, once on the HIP path and once with the CPU oracle standing in
    for the device call: unweighted counts bit-identical, weighted within 1e-10, correlation estimates equal."""
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine

    rng = np.random.default_rng(500 + seed)
    size = rng.choice([3.0, 12.0, 40.0])  # degrees
    ra0, dec0 = rng.uniform(0, 360), rng.uniform(-60, 60 - size)
    weighted = bool(seed % 2)

    def frame(n, with_z):
        d = {"ra": rng.uniform(ra0, ra0 + size, n),
             "dec": np.rad2deg(np.arcsin(rng.uniform(np.sin(np.deg2rad(dec0)), np.sin(np.deg2rad(dec0 + size)), n)))}
        if with_z:
            d["z"] = rng.uniform(0.05, 1.1, n)  # some objects fall outside the binning
        if weighted:
            d["w"] = rng.uniform(0.3, 2.0, n)
        return d

    num_patches = int(rng.integers(3, 9))
    kwargs = dict(ra_name="ra", dec_name="dec", weight_name="w" if weighted else None)
    ref = yaw.Catalog.from_dataframe(None, frame(6000, True), redshift_name="z", patch_num=num_patches, **kwargs)
    make = lambda n, with_z: yaw.Catalog.from_dataframe(None, frame(n, with_z), redshift_name="z" if with_z else None,
                                                        patch_centers=ref, **kwargs)
    unk, ref_rand, unk_rand = make(8000, False), make(12000, True), make(15000, False)
    num_scales = int(rng.integers(1, 4))
    if seed % 3 == 0:  # physical scales: the angle depends on the redshift bin
        lo = np.sort(rng.uniform(50.0, 400.0, num_scales)); hi = lo * rng.uniform(2.0, 6.0, num_scales); unit = "kpc"
    else:
        lo = np.sort(rng.uniform(0.02, 0.2, num_scales)) * size; hi = lo * rng.uniform(2.0, 5.0, num_scales); unit = "arcmin"
    config = yaw.Configuration.create(rmin=lo.tolist(), rmax=hi.tolist(), unit=unit, zmin=0.1, zmax=1.0,
                                      num_bins=int(rng.integers(2, 8)), closed="left" if seed % 2 else "right",
                                      rweight=-0.8 if seed in (1, 4) else None, resolution=12 if seed in (1, 4) else None)

    def measure():
        cross = yaw.crosscorrelate(config, ref, unk, ref_rand=ref_rand, unk_rand=unk_rand)
        auto = yaw.autocorrelate(config, ref, ref_rand)
        return cross, auto

    dev_cross, dev_auto = measure()
    helpers.use_oracle_engine(monkeypatch)
    ora_cross, ora_auto = measure()
    total = 0.0
    for dev, ora in ((dev_cross, ora_cross), (dev_auto, ora_auto)):
        assert len(dev) == len(ora) == num_scales
        for cd, co in zip(dev, ora):
            for kind in ("dd", "dr", "rd", "rr"):
                a, b = getattr(cd, kind), getattr(co, kind)
                assert (a is None) == (b is None)
                if a is None:
                    continue
                if weighted:
                    np.testing.assert_allclose(a.counts.counts, b.counts.counts, rtol=1e-10, atol=0)
                else:
                    assert np.array_equal(a.counts.counts, b.counts.counts), (seed, kind)
                assert np.array_equal(a.sum_weights.sum_weights1, b.sum_weights.sum_weights1)
                total += b.counts.counts.sum()
            np.testing.assert_allclose(cd.sample().data, co.sample().data, rtol=1e-9, atol=1e-12, equal_nan=True)
    assert total > 1000


@pytest.mark.parametrize("tag,cfg", [("w", "s2"), ("u", "rw"), ("w", "rw")])
def test_batch_submission_equals_single_calls_bit_for_bit(tag, cfg):
    """``yawhip_count_pairs_dense_batch`` (ABI 5; DD, DR, RD, RR of a cross-correlation and DD, DR, RR of an autocorrelation
    as ONE submission, src/yaw/correlation/measurements.py:617-628,517-523) returns what as many single
    ``yawhip_count_pairs_dense`` calls return -- every tensor bit for bit, weighted sums included -- with one fine bin per
    scale (values scattered from the slot's result block) and with separation weights (recombined on the device)."""
    import yet_another_wizz_amd as yaw

    inp, cats = helpers.full_catalogs(tag)
    config = helpers.full_config(inp, cfg, "right")
    ref, unk, rr, ur = cats["ref"], cats["unk"], cats["ref_rand"], cats["unk_rand"]
    for cat in (ref, rr):
        cat.build_trees(config.binning.edges, closed=config.binning.closed)
    for cat in (unk, ur):
        cat.build_trees(None)
    links = yaw.PatchLinkage.from_catalogs(config, ref, unk, rr, ur)
    requests = [((ref, unk), "DD"), ((ref, ur), "DR"), ((rr, unk), "RD"), ((rr, ur), "RR"), ((ref,), "DD"), ((rr,), "RR"),
                ((None, unk), "RD")]
    singles = [links.count_pairs(*cats_) if None not in cats_ else None for cats_, _ in requests]
    for _ in range(2):  # the second round runs on remembered plans
        batch = links.count_pairs_batch(requests)
        assert len(batch) == len(requests)
        for one, many in zip(singles, batch):
            if one is None:
                assert many == [None] * config.scales.num_scales
                continue
            for a, b in zip(one, many):
                assert np.array_equal(a.counts.counts, b.counts.counts)
                assert a.counts.counts.sum() > 0 and a.counts.auto == b.counts.auto
                assert np.array_equal(a.sum_weights.sum_weights1, b.sum_weights.sum_weights1)
    # more requests than slots in flight (four): the fifth waits for the first one's slot
    many = links.count_pairs_batch([((ref, unk), "DD")] * 6)
    for res in many:
        assert np.array_equal(res[0].counts.counts, singles[0][0].counts.counts)


def test_remembered_plans_follow_their_inputs():
    """The host side of a call (kernel choice, job and threshold tables on the device) is remembered per set of inputs:
    same inputs -> same result from the remembered plan; other thresholds, another job list, a changed option or a
    re-uploaded catalogue -> a fresh plan, never a stale one."""
    from yet_another_wizz_amd import _lib, engine

    rng = np.random.default_rng(77)
    P = 6

    def layout(n, bins):
        import yet_another_wizz_amd as yaw

        ra, dec = rng.uniform(40.0, 48.0, n), rng.uniform(-4.0, 4.0, n)
        centers = yaw.AngularCoordinates(np.deg2rad([[41.5 + 2.5 * (i % 3), -2.0 + 4.0 * (i // 3)] for i in range(P)]))
        z = rng.uniform(0.1, 0.9, n) if bins else None
        cat = yaw.Catalog.from_arrays(ra, dec, redshifts=z, patch_centers=centers)
        return cat.build_trees(np.linspace(0.1, 0.9, 5) if bins else None)

    l1, l2 = layout(30000, True), layout(40000, False)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    t1 = np.tile(np.array([[1e-7, 4e-6]]), (4, 1))
    t2 = np.tile(np.array([[2e-7, 9e-6]]), (4, 1))
    truth = {}
    for name, t, jb in (("t1", t1, jobs), ("t2", t2, jobs), ("half", t1, jobs[::2].copy())):
        truth[name], _ = engine.count_fine(l1, l2, jb, t, kernel="exact")
    for _ in range(3):  # alternating inputs: each finds its own plan
        for name, t, jb in (("t1", t1, jobs), ("t2", t2, jobs), ("half", t1, jobs[::2].copy())):
            got, st = engine.count_fine(l1, l2, jb, t)
            assert np.array_equal(got, truth[name]), name
            assert st.kernel_used == _lib.KERNEL_BAND
    # thresholds changed IN PLACE behind the same address: compared by content, not by pointer
    t3 = t1.copy()
    a, _ = engine.count_fine(l1, l2, jobs, t3)
    t3[:, 1] = t2[:, 1]
    t3[:, 0] = t2[:, 0]
    b, _ = engine.count_fine(l1, l2, jobs, t3)
    assert np.array_equal(a, truth["t1"]) and np.array_equal(b, truth["t2"])
    # an option changes the decisions: plans made before it are not used
    ctx = engine.get_context()
    ctx.set_option("band_fp32", 0)
    c, st = engine.count_fine(l1, l2, jobs, t1)
    ctx.set_option("band_fp32", 1)
    assert st.band_variant == 64 and np.array_equal(c, truth["t1"])
    d, st = engine.count_fine(l1, l2, jobs, t1)
    assert st.band_variant == 32 and np.array_equal(d, truth["t1"])
    # a catalogue that is freed and uploaded again (possibly at the same address) gets new plans
    for _ in range(3):
        for lay in (l1, l2):
            for dev in list(lay.device.values()):
                dev.free()
            lay.device.clear()
        e, _ = engine.count_fine(l1, l2, jobs, t2)
        assert np.array_equal(e, truth["t2"])
    # more distinct inputs than plans kept (16): the oldest go, results stay right
    for i in range(20):
        ti = t1 * (1.0 + 0.01 * i)
        f, _ = engine.count_fine(l1, l2, jobs[: 6 + i], ti)
        g, _ = engine.count_fine(l1, l2, jobs[: 6 + i], ti, kernel="exact")
        assert np.array_equal(f, g), i
