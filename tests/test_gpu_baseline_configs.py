"""GPU: the BASELINE.json configurations at FULL size, checked against the reference itself.

``tests/golden/fullsize_reference_totals.json`` holds, per scale and redshift bin, the pair totals the reference's own
``PatchLinkage.count_pairs`` produced for exactly these inputs (``tools/time_reference.py``, build container; the inputs
are ``bench.py``'s recipe with fixed seeds, so they are regenerated here bit for bit), and
``tests/golden/fullsize_slots_*.npz`` the reference's values SLOT BY SLOT -- ``counts[scale][:, i, j]`` of every linked
patch pair (config #3, unweighted and weighted) or of every eighth one (configs #4, #5) plus its ``sum_weights`` -- so
a job that landed in the wrong ``[i, j]`` slot cannot hide in a total (src/yaw/correlation/measurements.py:354-364).
On top of that: the oracle on a few whole jobs, the other device code paths on samples of jobs, symmetry and run-to-run
reproducibility.

  config #3  10M x 10M, 30 z-bins, 64 patches, DD of a cross-correlation (unweighted: totals must match exactly)
  config #4  10M data + 100M randoms, per-object weights, 64 patches, autocorrelation DD / DR / RR (1e-10 relative)
  config #5  50M x 50M, 3 log-spaced scales, 128 patches (exact)
"""
import json
import os
import types

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "fullsize_reference_totals.json")) as f:
        return json.load(f)


def _slots(name):
    return np.load(os.path.join(GOLDEN, f"fullsize_slots_{name}.npz"))


def _check_slots(result, slots, name, exact):
    """result: list of NormalisedCounts per scale, as ``PatchLinkage.count_pairs`` returns them; slots: the reference's
    ``counts[s][:, i, j]`` for the patch pairs ``<name>_ids`` and its ``sum_weights``."""
    ids, values = slots[f"{name}_ids"], slots[f"{name}_values"]  # [n, 2], [n, S, B]
    assert values.shape[1] == len(result)
    for s, res in enumerate(result):
        got = res.counts.counts[:, ids[:, 0], ids[:, 1]].T  # [n, B]
        if exact:
            assert np.array_equal(got, values[:, s]), f"{name} scale {s}"
        else:
            np.testing.assert_allclose(got, values[:, s], rtol=1e-10, atol=0, err_msg=f"{name} scale {s}")
    assert values.sum() > 0
    sw = result[0].sum_weights
    for got, key in ((sw.sum_weights1, "sum_weights1"), (sw.sum_weights2, "sum_weights2")):
        if exact:
            assert np.array_equal(got, slots[f"{name}_{key}"]), key
        else:  # summation order of the weights differs from the reference's
            np.testing.assert_allclose(got, slots[f"{name}_{key}"], rtol=1e-12, atol=0, err_msg=key)


def _as_cat(layout):
    return dict(x=layout.x, y=layout.y, z=layout.z, w=layout.w, nb=layout.num_bins, off=layout.offsets)


def _per_scale_bin(plans_combine, fine, jobs, auto):
    """fine[J, B, E-1] -> totals [S, B] with the reference's halving of the auto diagonal."""
    per_scale = plans_combine(np.moveaxis(fine, 0, -1))  # [S, B, J]
    if auto:
        per_scale = per_scale * np.where(jobs[:, 0] == jobs[:, 1], 0.5, 1.0)
    return per_scale.sum(axis=2)


def _setup(config, c1, c2):
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd.measurements import CombinePlan, angular_plans, threshold_table

    links = yaw.PatchLinkage.from_catalogs(config, c1, c2)
    plans = angular_plans(config)
    return links, threshold_table(plans), CombinePlan(plans)


def test_config3_headline_totals_match_reference(golden):
    import bench
    from yet_another_wizz_amd import _lib, engine

    g = golden["config3"]
    args = types.SimpleNamespace(n_ref=g["n_ref"], n_unk=g["n_unk"], patches=g["patches"], zbins=g["z_bins"])
    config, ref, unk = bench.make_catalogs(args)
    lref = ref.build_trees(config.binning.edges, closed=config.binning.closed)
    lunk = unk.build_trees(None)
    links, t, combine = _setup(config, ref, unk)
    jobs = links.get_patch_pairs(ref, unk)
    assert len(jobs) == g["linked_patch_pairs"]
    fine, st = engine.count_fine(lref, lunk, jobs, t)
    assert st.kernel_used == _lib.KERNEL_BAND and st.layout_mode == 1 and st.n_orientations == 3
    assert st.candidate_pairs == g["candidate_pairs"]
    assert np.array_equal(_per_scale_bin(combine, fine, jobs, False), np.array(g["pairs_per_scale_bin"]["DD"]))
    # the public entry point scatters the same numbers -- and every one of the 440 x 30 slots holds the reference's value
    result = links.count_pairs(ref, unk)
    (counts,) = result
    assert np.array_equal(counts.counts.counts.sum(axis=(1, 2)), np.array(g["pairs_per_scale_bin"]["DD"][0]))
    slots = _slots("config3")
    assert slots["DD_ids"].shape == (440, 2) and slots["candidate_pairs"] == st.candidate_pairs
    _check_slots(result, slots, "DD", exact=True)
    assert np.count_nonzero(counts.counts.counts.sum(axis=0)) <= 440  # nothing outside the linked pairs
    # the float64 band kernel (band_fp32 = 0) on every 10th job: the same counts
    ctx = engine.get_context()
    ctx.set_option("band_fp32", 0)
    f64, st64 = engine.count_fine(lref, lunk, jobs[::10], t)
    ctx.set_option("band_fp32", 1)
    assert st64.kernel_used == _lib.KERNEL_BAND and np.array_equal(f64, fine[::10])
    # the other device paths on every 40th job, the oracle on one whole job (2.4e10 candidate pairs)
    sample = jobs[::40]
    for kernel in ("sweep", "filter"):
        f2, _ = engine.count_fine(lref, lunk, sample, t, kernel=kernel)
        assert np.array_equal(f2, fine[::40]), kernel
    exp, _ = oracle.count_jobs(_as_cat(lref), _as_cat(lunk), jobs[:1], t)
    assert np.array_equal(fine[:1], exp.astype(np.float64))
    ref.drop_layouts()
    unk.drop_layouts()


def test_config5_three_scales_128_patches(golden):
    """50M x 50M, scales 0.5-1.58, 1.58-5, 5-15.8 arcmin (E = 4 distinct edges -> 3 fine bins per z-bin)."""
    import bench
    from yet_another_wizz_amd import _lib, engine

    g = golden["config5"]
    args = types.SimpleNamespace(n_ref=g["n_ref"], n_unk=g["n_unk"], patches=g["patches"], zbins=g["z_bins"], scales=3)
    config, ref, unk = bench.make_catalogs(args)
    lref = ref.build_trees(config.binning.edges, closed=config.binning.closed)
    lunk = unk.build_trees(None)
    links, t, combine = _setup(config, ref, unk)
    jobs = links.get_patch_pairs(ref, unk)
    assert len(jobs) == g["linked_patch_pairs"] and t.shape == (30, 4)
    fine, st = engine.count_fine(lref, lunk, jobs, t)
    assert st.kernel_used == _lib.KERNEL_BAND and st.layout_mode == 1
    assert st.candidate_pairs == g["candidate_pairs"]
    totals = _per_scale_bin(combine, fine, jobs, False)
    assert totals.shape == (3, 30) and np.array_equal(totals, np.array(g["pairs_per_scale_bin"]["DD"]))
    again, _ = engine.count_fine(lref, lunk, jobs, t)
    assert np.array_equal(again, fine)
    _check_slots(links.count_pairs(ref, unk), _slots("config5"), "DD", exact=True)  # every 8th linked pair, 3 scales x 30 bins
    sample = jobs[::60]
    f2, _ = engine.count_fine(lref, lunk, sample, t, kernel="sweep")
    assert np.array_equal(f2, fine[::60])
    f3, _ = engine.count_fine(lref, lunk, sample[:4], t, kernel="filter")
    assert np.array_equal(f3, fine[::60][:4])
    # the oracle on one diagonal job (1.5e11 candidate pairs)
    exp, _ = oracle.count_jobs(_as_cat(lref), _as_cat(lunk), jobs[:1], t)
    assert np.array_equal(fine[:1], exp.astype(np.float64))
    ref.drop_layouts()
    unk.drop_layouts()


def test_config4_weighted_autocorrelation_with_10x_randoms(golden):
    """10M data + 100M randoms, w ~ U(0.5, 1.5): DD, DR, RR on the natural path of each count (RR runs on the
    per-(patch, bin) strip layouts at their default density threshold)."""
    import bench
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import _lib, engine

    g = golden["config4"]
    config, data, rand = bench.make_auto_catalogs(g["n_ref"], g["n_unk"], weighted=True, patches=g["patches"])
    ld = data.build_trees(config.binning.edges, closed=config.binning.closed)
    lr = rand.build_trees(config.binning.edges, closed=config.binning.closed)
    links, t, combine = _setup(config, data, rand)
    modes, kernels = {}, {}
    for name, l1, l2, auto in (("DD", ld, ld, True), ("DR", ld, lr, False), ("RR", lr, lr, True)):
        jobs = links.get_patch_pairs(data, None if auto else rand)
        fine, st = engine.count_fine(l1, l2, jobs, t)
        modes[name], kernels[name] = st.layout_mode, st.kernel_used
        exp_tot = np.array(g["pairs_per_scale_bin"][name])
        np.testing.assert_allclose(_per_scale_bin(combine, fine, jobs, auto), exp_tot, rtol=1e-10, atol=0, err_msg=name)
        again, _ = engine.count_fine(l1, l2, jobs, t)
        assert np.array_equal(again, fine), name  # run-to-run bit reproducibility of the weighted sums
        # the oracle on whole jobs: a diagonal one and a neighbour pair (brute force over every same-bin pair)
        pick = np.array([0, len(jobs) - 1])
        exp_c, exp_s = oracle.count_jobs(_as_cat(l1), _as_cat(l2), jobs[pick], t)
        np.testing.assert_allclose(fine[pick], exp_s, rtol=1e-10, atol=0, err_msg=name)
        assert exp_c.sum() > 0
        # the pre-filter path on a sample of jobs (another summation order)
        sample = jobs[::60]
        f2, _ = engine.count_fine(l1, l2, sample, t, kernel="filter")
        np.testing.assert_allclose(f2, fine[::60], rtol=1e-12, atol=0, err_msg=name)
        if auto:  # ordered pairs: counts[p, q] == counts[q, p] is implied by summing i <= j only; the diagonal of the
            # integer counts is even
            ctx = engine.get_context()
            d1 = engine.device_catalog(l1, ctx)
            counts, _, _ = _lib.count_pairs(ctx, d1, d1, jobs[jobs[:, 0] == jobs[:, 1]][:8], t, want_counts=True, want_sums=False)
            assert np.all(counts % 2 == 0) and counts.sum() > 0
    # every count runs on the per-(patch, bin) strip layouts, on the band kernel (float32 classification: its fixed cost per
    # item is below the sweep kernel's also where the streamed runs hold a few dozen objects)
    assert modes == {"DD": 3, "DR": 3, "RR": 3}
    assert kernels["RR"] == kernels["DD"] == kernels["DR"] == _lib.KERNEL_BAND
    # the public entry point end to end (catalogues resident): Landy-Szalay amplitudes are finite and small
    (cf,) = yaw.autocorrelate(config, data, rand)
    assert cf.dd is not None and cf.dr is not None and cf.rr is not None
    # slot by slot against the reference (every 8th job of DD, DR, RR; the diagonal jobs carry the reference's x 0.5)
    slots = _slots("config4")
    for name, res in (("DD", cf.dd), ("DR", cf.dr), ("RR", cf.rr)):
        _check_slots([res], slots, name, exact=False)
    w = cf.sample().data
    assert np.all(np.isfinite(w)) and np.all(np.abs(w) < 0.05)  # uniform sky: no clustering
    data.drop_layouts()
    rand.drop_layouts()
