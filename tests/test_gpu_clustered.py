"""GPU: a strongly clustered survey (tests/clustered_sky.py: 3M reference x 4M unknown objects, 70 % of them in 300
clumps of 0.02-0.8 degrees, 24 patches of very different sizes, 12 z-bins, two scales sharing an edge) against the
REFERENCE's own pair-count tensors for exactly these columns (tools/make_golden_clustered.py ->
tests/golden/clustered_reference_counts.npz). Uniform skies keep every band of the band kernel near its mean length;
here windows of thousands of entries (many LDS stages) sit next to windows of a handful, and every device code path
has to agree on them."""
import os

import numpy as np
import pytest

import clustered_sky as cs
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import yet_another_wizz_amd as yaw

    g = np.load(os.path.join(GOLDEN, "clustered_reference_counts.npz"))
    centers = yaw.AngularCoordinates(cs.patch_centers())
    r, u = cs.sample(101, int(g["n_ref"]), with_z=True), cs.sample(202, int(g["n_unk"]), with_z=False, with_w=True)
    ref = yaw.Catalog.from_arrays(r["ra"], r["dec"], redshifts=r["z"], patch_centers=centers, degrees=False)
    unk = yaw.Catalog.from_arrays(u["ra"], u["dec"], weights=u["w"], patch_centers=centers, degrees=False)
    rmin, rmax = cs.SCALES_ARCMIN
    config = yaw.Configuration.create(rmin=rmin, rmax=rmax, unit="arcmin", edges=cs.bin_edges())
    ref.build_trees(config.binning.edges, closed=config.binning.closed)
    unk.build_trees(None)
    return g, config, ref, unk


def test_patches_are_the_references(setup):
    g, _, ref, unk = setup
    assert np.array_equal(np.asarray(ref.get_num_records()), g["num_records_ref"])
    assert np.array_equal(np.asarray(unk.get_num_records()), g["num_records_unk"])
    sizes = np.asarray(ref.get_num_records())
    assert sizes.max() > 8 * sizes.min()  # the patches really differ in size


@pytest.mark.parametrize("kernel", ["auto", "band", "sweep"])
def test_cross_counts_match_the_reference(setup, kernel):
    """reference (binned, unweighted) x unknown (unbinned, weighted), slot by slot against the reference's tensor.

    Tolerance. For slots of ~1e9 weighted pairs the REFERENCE is not exactly rounded: scipy adds products of node weight
    sums to running totals and ends up to 7e-10 (relative) away from the exact sum -- shown by recomputing the twelve
    largest slots from integer pair counts per unknown object and math.fsum (``cross_exact_*`` in the golden file,
    tools/make_golden_clustered.py). Against those exact values this path is held to 1e-12; against the reference's own
    numbers to 2e-9 where a slot holds more than 1e8 pairs and to the contract's 1e-10 everywhere else."""
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine

    g, config, ref, unk = setup
    links = yaw.PatchLinkage.from_catalogs(config, ref, unk)
    old = engine.default_kernel
    engine.default_kernel = kernel
    try:
        res = links.count_pairs(ref, unk)
    finally:
        engine.default_kernel = old
    got = np.stack([r.counts.counts for r in res])
    exp = g["cross_counts"]
    assert got.shape == exp.shape
    assert np.array_equal(got == 0, exp == 0)  # the same slots are filled
    big = exp > 1e8
    np.testing.assert_allclose(got[~big], exp[~big], rtol=1e-10, atol=0)
    np.testing.assert_allclose(got[big], exp[big], rtol=2e-9, atol=0)
    idx = g["cross_exact_idx"]
    np.testing.assert_allclose(got[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]], g["cross_exact_val"], rtol=1e-12, atol=0)
    ref_err = np.abs(exp[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] / g["cross_exact_val"] - 1.0).max()
    assert 1e-10 < ref_err < 2e-9  # the reference's own rounding at this size (what the looser bound above is for)
    np.testing.assert_allclose(res[0].sum_weights.sum_weights1, g["cross_sum_weights1"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(res[0].sum_weights.sum_weights2, g["cross_sum_weights2"], rtol=1e-12, atol=0)
    if kernel == "auto":
        assert links.last_stats.evaluated_pairs < 0.02 * links.last_stats.candidate_pairs


@pytest.mark.parametrize("kernel", ["auto", "band", "sweep"])
def test_auto_counts_match_the_reference_exactly(setup, kernel):
    """Autocorrelation count of the (unweighted) reference sample: integers, bit-identical to the reference's tensor
    including the halved diagonal."""
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import engine

    g, config, ref, _ = setup
    links = yaw.PatchLinkage.from_catalogs(config, ref)
    old = engine.default_kernel
    engine.default_kernel = kernel
    try:
        res = links.count_pairs(ref)
    finally:
        engine.default_kernel = old
    got = np.stack([r.counts.counts for r in res])
    assert np.array_equal(got, g["auto_counts"])
    assert float(got.sum()) > 5e9  # billions of pairs: dense clumps, long bands


def test_full_measurement_physical_scales_separation_weights_randoms():
    """A complete ``crosscorrelate`` at mid size on the clustered survey, as a user of the reference would run it:
    physical scales 150-1500 kpc (every z-bin its own thresholds), ``rweight=-1`` with 30 log bins (31 fine bins per
    z-bin), weighted unknown sample, two random samples -> DD, DR, RD, RR tensors, patch sums and ``CorrFunc.sample()``
    (w(z), jackknife samples, errors, covariance) against the reference's own outputs
    (tools/make_golden_clustered.py --full; D_A(z) of the reference's cosmology stand-in is this package's)."""
    import yet_another_wizz_amd as yaw
    from helpers import check_corrfuncs
    from make_golden_clustered_params import FULL

    g = np.load(os.path.join(GOLDEN, "clustered_reference_full.npz"))
    centers = yaw.AngularCoordinates(cs.patch_centers())

    def cat(seed, n, frac, with_z, with_w):
        c = cs.sample(seed, int(n), clustered_fraction=frac, with_z=with_z, with_w=with_w)
        return yaw.Catalog.from_arrays(c["ra"], c["dec"], redshifts=c.get("z"), weights=c.get("w"), patch_centers=centers,
                                       degrees=False)

    ref, unk = cat(101, FULL["n_ref"], 0.7, True, False), cat(202, FULL["n_unk"], 0.7, False, True)
    ref_rand, unk_rand = cat(303, FULL["n_ref_rand"], 0.0, True, False), cat(404, FULL["n_unk_rand"], 0.0, False, True)
    config = yaw.Configuration.create(rmin=FULL["rmin"], rmax=FULL["rmax"], unit=FULL["unit"], rweight=FULL["rweight"],
                                      resolution=FULL["resolution"], edges=cs.bin_edges())
    cfs = yaw.crosscorrelate(config, ref, unk, ref_rand=ref_rand, unk_rand=unk_rand)
    check_corrfuncs("cross", cfs, g, exact=lambda kind: False)
    assert np.all(np.isfinite(cfs[0].sample().data)) and cfs[0].sample().data.max() > 50  # a strongly clustered sample
    # and the autocorrelation of the reference sample against its randoms (binned x binned: DD, DR, RR)
    check_corrfuncs("auto", yaw.autocorrelate(config, ref, ref_rand), g, exact=lambda kind: False)
