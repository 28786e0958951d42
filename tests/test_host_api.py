"""CPU tests of the host layer: the reference's API shapes, its error behaviour, and -- with the
device call replaced by the CPU oracle -- the whole crosscorrelate / autocorrelate driver against
outputs captured from the reference (tools/make_golden.py)."""
import numpy as np
import pytest

import helpers
import yet_another_wizz_amd as yaw
from yet_another_wizz_amd import angular_bins, engine


@pytest.fixture
def oracle_engine(monkeypatch):
    helpers.use_oracle_engine(monkeypatch)


@pytest.mark.parametrize("tag", ["u", "w"])
@pytest.mark.parametrize("cfg,closed", [("s2", "right"), ("s2", "left"), ("rw", "right")])
def test_full_driver_vs_reference(oracle_engine, tag, cfg, closed):
    helpers.run_full_case(tag, cfg, closed)


def test_twodflens_vs_reference(oracle_engine):
    helpers.run_twodflens_case()


def test_reference_end_to_end_known_answer(oracle_engine):
    helpers.run_reference_example_case()


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: without a GPU the real engine must raise, never return numbers."""
    from yet_another_wizz_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    inp, cats = helpers.full_catalogs("u")
    config = helpers.full_config(inp, "s2", "right")
    with pytest.raises(_lib.YawhipError):
        yaw.crosscorrelate(config, cats["ref"], cats["unk"], unk_rand=cats["unk_rand"])


def test_crosscorrelate_errors(oracle_engine):
    inp, cats = helpers.full_catalogs("u")
    config = helpers.full_config(inp, "s2", "right")
    with pytest.raises(ValueError, match="at least one random dataset"):  # measurements.py:588-589
        yaw.crosscorrelate(config, cats["ref"], cats["unk"])
    with pytest.raises(ValueError, match="no 'redshifts'"):  # trees.py:396-397
        yaw.crosscorrelate(config, cats["unk"], cats["ref"], unk_rand=cats["unk_rand"])
    with pytest.raises(ValueError, match="separate"):
        yaw.crosscorrelate(config, cats["ref"], cats["ref"], unk_rand=cats["unk_rand"])
    # mismatching patch ids (measurements.py:210-211)
    sub = {k: inp[f"unk.{k}"] for k in ("ra", "dec")}
    other = yaw.Catalog.from_dataframe(None, sub, ra_name="ra", dec_name="dec",
                                       patch_centers=yaw.AngularCoordinates(inp["patch_centers"][:4]))
    with pytest.raises(yaw.InconsistentPatchesError, match="patch IDs do not match"):
        yaw.crosscorrelate(config, cats["ref"], other, unk_rand=cats["unk_rand"])
    # same number of patches but shuffled centres (measurements.py:148-149)
    shuffled = yaw.Catalog.from_dataframe(None, sub, ra_name="ra", dec_name="dec",
                                          patch_centers=yaw.AngularCoordinates(inp["patch_centers"][::-1].copy()))
    with pytest.raises(yaw.InconsistentPatchesError, match="not aligned"):
        yaw.crosscorrelate(config, cats["ref"], shuffled, unk_rand=cats["unk_rand"])
    with pytest.raises(FileNotFoundError):
        fresh = yaw.Catalog.from_dataframe(None, sub, ra_name="ra", dec_name="dec",
                                           patch_centers=yaw.AngularCoordinates(inp["patch_centers"]))
        yaw.PatchLinkage.from_catalogs(config, cats["ref"], fresh).count_pairs(cats["ref"], fresh)


def test_davis_peebles_when_no_rr(oracle_engine):
    inp, cats = helpers.full_catalogs("u")
    config = helpers.full_config(inp, "s2", "right")
    cfs = yaw.crosscorrelate(config, cats["ref"], cats["unk"], unk_rand=cats["unk_rand"])
    assert len(cfs) == 2 and cfs[0].rr is None and cfs[0].rd is None and cfs[0].dr is not None
    assert cfs[0].get_estimator().name == "DP"
    exp = helpers.load_golden("full_u_s2_right.npz")
    assert np.array_equal(cfs[1].dr.counts.counts, exp["cross.s1.dr.counts"])
    cfs = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"], count_rr=False)
    assert cfs[0].rr is None and cfs[0].auto and cfs[0].dd.auto and not cfs[0].dr.auto


# ---------------------------------------------------------------- helpers mirrored from the reference's own tests
@pytest.mark.parametrize("ang_min,ang_max", [(0.0, np.pi), ([0.0, 3.0], [0.1, np.pi]), ([0.0, 1.0], [1.0, np.pi])])
def test_parse_ang_limits(ang_min, ang_max):  # reference tests/catalog/test_trees.py:14-26
    assert np.array_equal(angular_bins.parse_ang_limits(ang_min, ang_max), np.column_stack((ang_min, ang_max)))


def test_parse_ang_limits_errors():  # test_trees.py:29-53
    with pytest.raises(ValueError, match=".*1-dim.*"):
        angular_bins.parse_ang_limits([[1.0]], [[10.0]])
    with pytest.raises(ValueError, match=".*length.*"):
        angular_bins.parse_ang_limits([1.0], [10.0, 100.0])
    with pytest.raises(ValueError, match=".*<.*"):
        angular_bins.parse_ang_limits([0.0, 0.1, 0.2], [0.2, 0.2, 0.2])
    for lo, hi in [(-0.01, 1.0), (0.0, np.pi + 1e9)]:
        with pytest.raises(ValueError, match=".*not in range.*"):
            angular_bins.parse_ang_limits(lo, hi)


@pytest.mark.parametrize(
    "rng,expect",
    [([[1.0, 10.0], [10.0, 100.0]], [1.0, 10.0, 100.0]), ([[1.0, 10.0], [11.0, 100.0]], [1.0, 10.0, 11.0, 100.0]),
     ([[1.0, 11.0], [10.0, 100.0]], [1.0, 10.0, 11.0, 100.0])],
)
def test_get_ang_bins(rng, expect):  # test_trees.py:56-67
    np.testing.assert_almost_equal(angular_bins.get_ang_bins(np.array(rng), None, None), expect)


@pytest.mark.parametrize(
    "rng,expect",
    [([[0.1, 9.0], [9.0, 1000.0]], [0.1, 1.0, 9.0, 10.0, 100.0, 1000.0]),
     ([[0.1, 10.0], [10.0, 1000.0]], [0.1, 1.0, 10.0, 100.0, 1000.0]),
     ([[0.1, 10.0], [11.0, 1000.0]], [0.1, 1.0, 10.0, 11.0, 100.0, 1000.0]),
     ([[0.1, 11.0], [10.0, 1000.0]], [0.1, 1.0, 10.0, 11.0, 100.0, 1000.0])],
)
def test_get_ang_bins_weights(rng, expect):  # test_trees.py:70-86
    np.testing.assert_almost_equal(angular_bins.get_ang_bins(np.array(rng), 1.0, 4), expect)


def test_logarithmic_mid():  # test_trees.py:89-93
    assert angular_bins.logarithmic_mid([1.0, 100.0]) == 10.0
    assert np.array_equal(angular_bins.logarithmic_mid([1.0, 100.0, 10000.0]), [10.0, 1000.0])


@pytest.mark.parametrize(
    "lims,expect",
    [([[1.0, 1000.0]], [3]), ([[10.0, 1000.0]], [2]), ([[1.0, 10.0], [100.0, 1000.0]], [1, 1]),
     ([[1.0, 10.0], [10.0, 1000.0]], [1, 2])],
)
def test_get_counts_for_limits(lims, expect):  # test_trees.py:107-131
    ang_bins = np.array([1.0, 10.0, 100.0, 1000.0])
    assert np.array_equal(angular_bins.get_counts_for_limits(np.ones(3), ang_bins, np.array(lims)), expect)


def test_plan_matches_reference_thresholds():
    sj = helpers.load_golden("single_job.npz")
    for key in (str(k) for k in sj["case_names"]):
        parts = key.split(".")
        lo, hi = sj[f"scales.{parts[0]}"]
        rwname = ".".join(parts[2:-1])
        rw, res = (None, None) if rwname == "plain" else (float(rwname[2:].rsplit("_", 1)[0]), int(rwname.rsplit("_", 1)[1]))
        plan = angular_bins.plan_for_limits(lo * np.pi / 10800, hi * np.pi / 10800, rw, res)
        assert np.array_equal(plan.ang_bins, sj[key + ".ang_bins"]), key
        assert np.array_equal(plan.thresholds, sj[key + ".t"]), key
        got = plan.combine(sj[key + ".fine"])
        assert np.array_equal(got, sj[key + ".final"]), key


# ---------------------------------------------------------------- coordinates (reference tests/test_coordinates.py)
def test_coordinates_roundtrip_and_fixed_points():
    pts = yaw.AngularCoordinates(np.deg2rad([[0.0, 0.0], [90.0, 0.0], [180.0, 0.0], [270.0, 0.0], [0.0, 90.0], [0.0, -90.0]]))
    expect = np.array([[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=float)
    np.testing.assert_allclose(pts.to_3d(), expect, atol=1e-15)
    back = yaw.AngularCoordinates.from_3d(expect)
    np.testing.assert_allclose(back.data, pts.data, atol=1e-15)
    d = yaw.AngularDistances.from_3d([0.0, np.sqrt(2.0), 2.0])
    np.testing.assert_allclose(d.data, [0.0, np.pi / 2, np.pi])
    np.testing.assert_allclose(d.to_3d(), [0.0, np.sqrt(2.0), 2.0])
    with pytest.raises(ValueError):
        yaw.AngularDistances.from_3d([2.1])
    gc = helpers.load_golden("greatcircle.npz")
    assert np.array_equal(yaw.AngularCoordinates(gc["radec"]).to_3d(), gc["xyz"])
    assert yaw.AngularCoordinates(gc["radec"][:7]).distance(yaw.AngularCoordinates(gc["radec"][0])).data[0] == 0.0
    m = yaw.AngularCoordinates(np.deg2rad([[10.0, 0.0], [30.0, 0.0]])).mean()
    np.testing.assert_allclose(np.rad2deg(m.data), [[20.0, 0.0]], atol=1e-12)


def test_binning_and_config():
    b = yaw.Binning([0.1, 0.2, 0.4, 0.8], closed="left")
    assert len(b) == 3 and np.allclose(b.mids, [0.15, 0.3, 0.6]) and np.allclose(b.dz, [0.1, 0.2, 0.4])
    assert np.array_equal(b.assign(np.array([0.05, 0.1, 0.2, 0.8, 0.79])), [-1, 0, 1, -1, 2])
    r = yaw.Binning([0.1, 0.2, 0.4, 0.8])
    assert np.array_equal(r.assign(np.array([0.05, 0.1, 0.2, 0.8, 0.81])), [-1, -1, 0, 2, -1])
    assert b[1:] == yaw.Binning([0.2, 0.4, 0.8], closed="left") and b != r
    with pytest.raises(ValueError):
        yaw.Binning([0.1, 0.1])
    c = yaw.Configuration.create(rmin=[100, 500], rmax=[1000, 1500], zmin=0.1, zmax=1.0, num_bins=9)
    assert c.scales.num_scales == 2 and c.binning.num_bins == 9 and c.scales.unit == "kpc"
    assert c == yaw.Configuration.from_dict(c.to_dict())
    assert c.modify(num_bins=3).binning.num_bins == 3
    lo, hi = c.scales.scales.get_angle_radian(0.5, c.cosmology)
    d_a = c.cosmology.angular_diameter_distance(0.5)
    np.testing.assert_allclose(lo, np.array([0.1, 0.5]) / d_a)
    np.testing.assert_allclose(hi, np.array([1.0, 1.5]) / d_a)
    from yet_another_wizz_amd.config import ConfigError

    with pytest.raises(ConfigError):
        yaw.Configuration.create(rmin=10, rmax=1, zmin=0.1, zmax=0.2)
    with pytest.raises(ConfigError):
        yaw.Configuration.create(rmin=1, rmax=10)


def test_catalog_constructors():
    rng = np.random.default_rng(3)
    n = 5000
    frame = dict(ra=rng.uniform(0, 40, n), dec=rng.uniform(-10, 10, n), z=rng.uniform(0, 1, n), w=rng.uniform(1, 2, n))
    cat = yaw.Catalog.from_dataframe(None, frame, ra_name="ra", dec_name="dec", redshift_name="z", weight_name="w", patch_num=6)
    assert len(cat) == cat.num_patches == 6 and list(cat) == list(range(6))
    assert cat.has_weights and cat.has_redshifts and sum(cat.get_num_records()) == n
    np.testing.assert_allclose(sum(cat.get_sum_weights()), frame["w"].sum())
    other = yaw.Catalog.from_dataframe(None, frame, ra_name="ra", dec_name="dec", patch_centers=cat)
    assert other.get_num_records() == cat.get_num_records() and not other.has_redshifts
    lay = cat.build_trees([0.2, 0.5, 0.9])
    assert lay.num_bins == 2 and lay.offsets[-1] == np.sum((frame["z"] > 0.2) & (frame["z"] <= 0.9))
    assert lay.sum_weights.shape == (2, 6)
    with pytest.raises(ValueError, match="no 'redshifts'"):
        other.build_trees([0.2, 0.5])
    with pytest.raises(ValueError, match="no patch method"):
        yaw.Catalog.from_dataframe(None, frame, ra_name="ra", dec_name="dec")
    with pytest.raises(ValueError, match="empty patches"):
        yaw.Catalog.from_arrays(frame["ra"], frame["dec"], patch_ids=np.where(np.arange(n) % 2, 0, 2))
    with pytest.raises(TypeError):
        yaw.Catalog.from_dataframe(None, frame, ra_name="ra", dec_name="dec", patch_centers=np.zeros((3, 2)))


def test_sort_axis_follows_the_footprint():
    from yet_another_wizz_amd.measurements import best_sort_axis

    ones = np.ones(3)
    assert best_sort_axis(np.array([[0.0, 0.1, 0.99], [0.1, 0.0, 0.99], [0.05, 0.05, 0.99]]), ones) in (0, 1)  # polar cap
    assert best_sort_axis(np.array([[0.99, 0.1, 0.0], [0.98, 0.0, 0.2], [0.97, 0.2, 0.1]]), ones) in (1, 2)   # around +x
    full = np.array([[1.0, 0, 0], [-1.0, 0, 0], [0, 1.0, 0], [0, -1.0, 0], [0, 0, 1.0], [0, 0, -1.0]])
    assert best_sort_axis(full, np.ones(6)) == 2  # no preferred direction: keep z


def test_strip_spacing_rule_and_small_helpers():
    """Pure host helpers around the device path: grid spacing chosen from the thresholds, the radix argsort used
    for patch / segment keys, and the per-configuration cache of angular plans."""
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import catalog, engine, measurements

    chord = lambda arcmin: (2.0 * np.sin(arcmin * np.pi / 10800 / 2.0)) ** 2
    assert engine.strip_micro_for(np.array([[chord(1.0), chord(10.0)]])) == 3000     # just above chord(10') = 2909
    assert engine.strip_micro_for(np.array([[chord(0.1), chord(0.5)]])) == 1000      # lower clamp
    assert engine.strip_micro_for(np.array([[chord(60.0), chord(3000.0)]])) == 100000  # upper clamp
    rng = np.random.default_rng(3)
    for num in (7, 70000):  # uint16 radix path and the general path
        keys = rng.integers(0, num, 5000)
        assert np.array_equal(catalog._stable_argsort_small(keys, num), np.argsort(keys, kind="stable"))
    config = yaw.Configuration.create(rmin=1.0, rmax=10.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=5)
    plans, t, _ = measurements._plans_for(config)
    assert measurements._plans_for(config)[1] is t and t.shape == (5, 2)
    other = yaw.Configuration.create(rmin=2.0, rmax=10.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=5)
    assert measurements._plans_for(other)[1] is not t


def test_device_helpers_fall_back_or_fail_as_documented(monkeypatch):
    """Without a GPU: patch assignment (catalogue preparation) quietly uses scipy, the pair-count entry
    points raise."""
    from yet_another_wizz_amd import _lib, catalog, engine

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    rng = np.random.default_rng(4)
    v = rng.normal(size=(1000, 3)); v /= np.linalg.norm(v, axis=1)[:, None]
    c = rng.normal(size=(5, 3)); c /= np.linalg.norm(c, axis=1)[:, None]
    assert engine.assign_patches(v, c) is None
    monkeypatch.setattr(catalog, "DEVICE_ASSIGN_MIN", 10)
    ids = catalog.nearest_center(v, c)
    assert np.array_equal(ids, ((v[:, None, :] - c[None, :, :]) ** 2).sum(axis=2).argmin(axis=1))
    with pytest.raises(_lib.YawhipError):
        engine.get_context()
