"""Shared helpers for the host-level tests (CPU and GPU variants compare against the same golden files)."""
import numpy as np

from conftest import load_golden
from oracle import oracle

RTOL_W = 1e-10


def oracle_count_fine(layout1, layout2, jobs, thresholds, *, kernel=None, sort_axis=2, max_workers=None):
    """Stand-in for yet_another_wizz_amd.engine.count_fine built on the CPU oracle: lets the CPU
    suite exercise the host driver (linkage, thresholds, recombination, sharding) without a GPU."""
    from yet_another_wizz_amd._lib import CountStats

    def as_cat(layout):
        return dict(x=layout.x, y=layout.y, z=layout.z, w=layout.w, nb=layout.num_bins, off=layout.offsets)

    counts, sums = oracle.count_jobs(as_cat(layout1), as_cat(layout2), jobs, thresholds)
    weighted = layout1.w is not None or layout2.w is not None
    return (sums if weighted else counts.astype(np.float64)), CountStats(candidate_pairs=0)


def oracle_count_dense(layout1, layout2, jobs, thresholds, slices, fine_factors, halve_diagonal, *, kernel=None, sort_axis=2,
                       max_workers=None):
    """Stand-in for yet_another_wizz_amd.engine.count_dense (``yawhip_count_pairs_dense``): the oracle's fine counts, then
    the host epilogue of ``PatchLinkage.count_pairs`` restated in numpy (separation weights, per-scale sums of the fine
    bins, x 0.5 on the diagonal jobs of an autocorrelation, scatter into [S, B, P, P])."""
    fine, stats = oracle_count_fine(layout1, layout2, jobs, thresholds)
    jobs = np.asarray(jobs).reshape(-1, 2)
    n_bins, n_scales, n_patches = fine.shape[1], slices.shape[1], layout1.num_patches
    weighted = fine if fine_factors is None else fine * fine_factors[np.newaxis]
    factor = np.where(jobs[:, 0] == jobs[:, 1], 0.5, 1.0) if halve_diagonal else 1.0
    dense = np.zeros((n_scales, n_bins, n_patches, n_patches), dtype=np.float64)
    for k in range(n_bins):
        for s in range(n_scales):
            lo, hi = slices[k, s]
            dense[s, k, jobs[:, 0], jobs[:, 1]] = weighted[:, k, lo:hi].sum(axis=1) * factor
    return dense, stats


def oracle_count_dense_batch(pairs, thresholds, slices, fine_factors, *, kernel=None, sort_axis=2, max_workers=None):
    """Stand-in for yet_another_wizz_amd.engine.count_dense_batch (``yawhip_count_pairs_dense_batch``): the requests one
    after the other through the stand-in of the single call."""
    return [oracle_count_dense(l1, l2, jobs, thresholds, slices, fine_factors, halve) for l1, l2, jobs, halve in pairs]


def use_oracle_engine(monkeypatch):
    """Replace the seams between the host driver and the HIP library by their oracle stand-ins."""
    from yet_another_wizz_amd import engine

    monkeypatch.setattr(engine, "count_fine", oracle_count_fine)
    monkeypatch.setattr(engine, "count_dense", oracle_count_dense)
    monkeypatch.setattr(engine, "count_dense_batch", oracle_count_dense_batch)


def full_catalogs(tag):
    """The four catalogues of tests/golden/full_<tag>_inputs.npz."""
    import yet_another_wizz_amd as yaw

    inp = load_golden(f"full_{tag}_inputs.npz")
    centers = yaw.AngularCoordinates(inp["patch_centers"])
    cats = {}
    for name in ("ref", "unk", "ref_rand", "unk_rand"):
        frame = {k.split(".", 1)[1]: inp[k] for k in inp.files if k.startswith(name + ".") and ".meta." not in k}
        cats[name] = yaw.Catalog.from_dataframe(
            None, frame, ra_name="ra", dec_name="dec", weight_name="w" if "w" in frame else None,
            redshift_name="z" if "z" in frame else None, patch_centers=centers,
        )
    return inp, cats


def full_config(inp, cfg, closed):
    import yet_another_wizz_amd as yaw

    if cfg == "s2":
        kw = dict(rmin=[2.0, 5.0], rmax=[20.0, 40.0], unit="arcmin")
    else:
        kw = dict(rmin=[2.0], rmax=[30.0], unit="arcmin", rweight=-0.8, resolution=12)
    return yaw.Configuration.create(edges=inp["zedges"], closed=closed, **kw)


def check_corrfuncs(prefix, cfs, exp, *, exact):
    """Compare a list of CorrFunc with the arrays the reference produced."""
    for s, cf in enumerate(cfs):
        for kind in ("dd", "dr", "rd", "rr"):
            key = f"{prefix}.s{s}.{kind}.counts"
            nc = getattr(cf, kind)
            if key not in exp.files:
                assert nc is None, key
                continue
            if exact(kind):
                assert np.array_equal(nc.counts.counts, exp[key]), key  # bit-identical
            else:
                np.testing.assert_allclose(nc.counts.counts, exp[key], rtol=RTOL_W, atol=0, err_msg=key)
            np.testing.assert_allclose(nc.sum_weights.sum_weights1, exp[f"{prefix}.s{s}.{kind}.sum_weights1"], rtol=1e-13)
            np.testing.assert_allclose(nc.sum_weights.sum_weights2, exp[f"{prefix}.s{s}.{kind}.sum_weights2"], rtol=1e-13)
            sp = nc.sample_patch_sum()
            np.testing.assert_allclose(sp.data, exp[f"{prefix}.s{s}.{kind}.sample_data"], rtol=1e-10)
            np.testing.assert_allclose(sp.samples, exp[f"{prefix}.s{s}.{kind}.sample_samples"], rtol=1e-10)
        cd = cf.sample()
        np.testing.assert_allclose(cd.data, exp[f"{prefix}.s{s}.corr_data"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(cd.samples, exp[f"{prefix}.s{s}.corr_samples"], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(cd.error, exp[f"{prefix}.s{s}.corr_error"], rtol=1e-7, atol=1e-12)
        np.testing.assert_allclose(cd.covariance, exp[f"{prefix}.s{s}.corr_covariance"], rtol=1e-7, atol=1e-14)


def run_full_case(tag, cfg, closed):
    """crosscorrelate + autocorrelate on the golden inputs, checked against the reference's outputs."""
    import yet_another_wizz_amd as yaw

    inp, cats = full_catalogs(tag)
    exp = load_golden(f"full_{tag}_{cfg}_{closed}.npz")
    config = full_config(inp, cfg, closed)
    # patch metadata feeds the linkage
    for name, cat in cats.items():
        np.testing.assert_allclose(cat.get_centers().data, inp[f"{name}.meta.centers"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(cat.get_radii().data, inp[f"{name}.meta.radii"], rtol=1e-9)
        assert np.array_equal(np.array(cat.get_num_records()), inp[f"{name}.meta.num_records"])
        np.testing.assert_allclose(np.array(cat.get_sum_weights()), inp[f"{name}.meta.sum_weights"], rtol=1e-13)
    links = yaw.PatchLinkage.from_catalogs(config, cats["ref"], cats["unk"], cats["ref_rand"], cats["unk_rand"])
    assert np.array_equal(np.array(sorted(links.iter_patch_id_pairs(auto=False))), exp["cross.job_pairs"])
    assert np.array_equal(np.array(sorted(links.iter_patch_id_pairs(auto=True))), exp["auto.job_pairs"])
    plain = cfg != "rw"
    weighted = {"ref": tag == "w", "unk": tag == "w", "ref_rand": False, "unk_rand": tag == "w"}
    cross_sides = dict(dd=("ref", "unk"), dr=("ref", "unk_rand"), rd=("ref_rand", "unk"), rr=("ref_rand", "unk_rand"))
    auto_sides = dict(dd=("ref", "ref"), dr=("ref", "ref_rand"), rr=("ref_rand", "ref_rand"))
    cfs = yaw.crosscorrelate(config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"])
    check_corrfuncs("cross", cfs, exp, exact=lambda k: plain and not any(weighted[c] for c in cross_sides[k]))
    cfs = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"], count_rr=True)
    check_corrfuncs("auto", cfs, exp, exact=lambda k: plain and not any(weighted[c] for c in auto_sides[k]))
    for cat in cats.values():
        cat.drop_layouts()


def twodflens_catalogs():
    import yet_another_wizz_amd as yaw

    g = load_golden("twodflens.npz")
    kw = dict(ra_name="RA", dec_name="Dec", weight_name="wei", patch_name="patch")
    data = {c: g[f"data.{c}"] for c in ("RA", "Dec", "redshift", "wei", "patch")}
    rand = {c: g[f"rand.{c}"] for c in ("RA", "Dec", "redshift", "wei", "patch")}
    cats = dict(
        data=yaw.Catalog.from_dataframe(None, data, redshift_name="redshift", **kw),
        rand=yaw.Catalog.from_dataframe(None, rand, redshift_name="redshift", **kw),
        unk=yaw.Catalog.from_dataframe(None, data, **kw),
    )
    config = yaw.Configuration.create(rmin=1.0, rmax=10.0, unit="arcmin", zmin=0.15, zmax=0.7, num_bins=11)
    return g, cats, config


def run_twodflens_case():
    import yet_another_wizz_amd as yaw

    g, cats, config = twodflens_catalogs()
    assert np.array_equal(np.asarray(config.binning.binning.edges), g["zedges"])
    np.testing.assert_allclose(cats["data"].get_centers().data, g["data.meta.centers"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(cats["data"].get_radii().data, g["data.meta.radii"], rtol=1e-9)
    cfs = yaw.crosscorrelate(config, cats["data"], cats["unk"], ref_rand=cats["rand"])
    check_corrfuncs("cross", cfs, g, exact=lambda k: False)
    cfs = yaw.autocorrelate(config, cats["data"], cats["rand"], count_rr=True)
    check_corrfuncs("auto", cfs, g, exact=lambda k: False)
    for cat in cats.values():
        cat.drop_layouts()


def run_reference_example_case():
    """The reference's own end-to-end known answer (tests/test_setups.py:155-172): bundled 2dFLenS
    data (src/yaw/examples/*.pqt, copied as data files), default configuration in *physical* units
    (100-1000 kpc, Planck15) -> crosscorrelate + autocorrelate -> n(z) = w_sp / sqrt(dz^2 w_ss)
    must equal src/yaw/examples/estimate.{dat,smp} to 6 decimals (assert_array_almost_equal)."""
    import os

    import yet_another_wizz_amd as yaw
    from conftest import GOLDEN
    from yet_another_wizz_amd.redshifts import RedshiftData

    d = os.path.join(GOLDEN, "reference_example")
    kw = dict(ra_name="RA", dec_name="Dec", weight_name="wei", redshift_name="redshift", patch_name="patch")
    data = os.path.join(d, "2dflens_kidss_data.pqt")
    ref = yaw.Catalog.from_file(None, data, **kw)
    unk = yaw.Catalog.from_file(None, data, **kw)
    rnd = yaw.Catalog.from_file(None, os.path.join(d, "2dflens_kidss_rand_5x.pqt"), **kw)
    assert ref.num_patches == 11 and sum(ref.get_num_records()) == 21875 and sum(rnd.get_num_records()) == 109375
    config = yaw.Configuration.create(rmin=100, rmax=1000, zmin=0.15, zmax=0.7, num_bins=11)  # examples/__init__.py:271
    (cross,) = yaw.crosscorrelate(config, ref, unk, ref_rand=rnd)     # create_example_data.py:23-26
    (auto,) = yaw.autocorrelate(config, ref, rnd)                      # create_example_data.py:29
    assert cross.rd is not None and cross.dr is None and cross.get_estimator().name == "DP"
    assert auto.rr is not None and auto.get_estimator().name == "LS"
    nz = RedshiftData.from_corrfuncs(cross, auto)                      # create_example_data.py:36
    dat = np.loadtxt(os.path.join(d, "estimate.dat"))
    smp = np.loadtxt(os.path.join(d, "estimate.smp"))
    cov = np.loadtxt(os.path.join(d, "estimate.cov"))
    np.testing.assert_array_almost_equal(np.column_stack([nz.binning.left, nz.binning.right]), dat[:, :2])
    np.testing.assert_array_almost_equal(nz.data, dat[:, 2])           # 6 decimals, as the reference's test
    np.testing.assert_array_almost_equal(nz.error, dat[:, 3])
    np.testing.assert_array_almost_equal(nz.samples.T, smp[:, 2:])
    np.testing.assert_array_almost_equal(nz.covariance, cov)
    for cat in (ref, unk, rnd):
        cat.drop_layouts()


def run_refcache_case():
    """Measurements taken from the cache the REFERENCE wrote (tests/golden/refcache: data.bin / meta.yml /
    patch_ids.bin, patch.py:164-178), against what the reference itself measured from that cache
    (refcache_counts.npz, tools/make_golden.py --refcache-counts). Weighted: 1e-10."""
    import os

    import yet_another_wizz_amd as yaw
    from conftest import GOLDEN

    exp_in = load_golden("refcache_expect.npz")
    exp = load_golden("refcache_counts.npz")
    frame = {c: exp_in[f"input.{c}"] for c in ("ra", "dec", "z", "w")}
    centers = yaw.AngularCoordinates(exp_in["patch_centers"])
    kw = dict(ra_name="ra", dec_name="dec", weight_name="w", patch_centers=centers)
    config = yaw.Configuration.create(rmin=0.5, rmax=8.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=4)
    ref = yaw.Catalog(os.path.join(GOLDEN, "refcache"))
    unk = yaw.Catalog.from_dataframe(None, frame, **kw)
    rnd = yaw.Catalog.from_dataframe(None, frame, **kw)
    rnz = yaw.Catalog.from_dataframe(None, frame, redshift_name="z", **kw)
    check_corrfuncs("cross", yaw.crosscorrelate(config, ref, unk, unk_rand=rnd), exp, exact=lambda k: False)
    check_corrfuncs("auto", yaw.autocorrelate(config, ref, rnz, count_rr=True), exp, exact=lambda k: False)
    assert exp["cross.s0.dd.counts"].sum() > 0 and exp["auto.s0.rr.counts"].sum() > 0
