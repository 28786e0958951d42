"""CPU: the oracle against the reference's pair counts for the clustered survey (tests/clustered_sky.py,
tests/golden/clustered_reference_counts.npz) on a few whole patch pairs -- pins the oracle on non-uniform densities too
(the GPU test of the same data is tests/test_gpu_clustered.py)."""
import os

import numpy as np

import clustered_sky as cs
from conftest import GOLDEN
from oracle import oracle


def _as_cat(layout):
    return dict(x=layout.x, y=layout.y, z=layout.z, w=layout.w, nb=layout.num_bins, off=layout.offsets)


def test_oracle_matches_reference_on_clustered_patch_pairs():
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd.measurements import CombinePlan, angular_plans, threshold_table

    g = np.load(os.path.join(GOLDEN, "clustered_reference_counts.npz"))
    centers = yaw.AngularCoordinates(cs.patch_centers())
    r, u = cs.sample(101, int(g["n_ref"]), with_z=True), cs.sample(202, int(g["n_unk"]), with_z=False, with_w=True)
    ref = yaw.Catalog.from_arrays(r["ra"], r["dec"], redshifts=r["z"], patch_centers=centers, degrees=False)
    unk = yaw.Catalog.from_arrays(u["ra"], u["dec"], weights=u["w"], patch_centers=centers, degrees=False)
    assert np.array_equal(np.asarray(ref.get_num_records()), g["num_records_ref"])  # same patches as the reference
    assert np.array_equal(np.asarray(unk.get_num_records()), g["num_records_unk"])
    rmin, rmax = cs.SCALES_ARCMIN
    config = yaw.Configuration.create(rmin=rmin, rmax=rmax, unit="arcmin", edges=cs.bin_edges())
    lref = ref.build_trees(config.binning.edges, closed=config.binning.closed)
    lunk = unk.build_trees(None)
    plans = angular_plans(config)
    t, combine = threshold_table(plans), CombinePlan(plans)

    exp = g["cross_counts"]  # [S, B, P, P]
    filled = np.argwhere(exp.sum(axis=(0, 1)) > 0)
    cost = np.array([g["num_records_ref"][p] * g["num_records_unk"][q] for p, q in filled], dtype=np.float64)
    order = np.argsort(cost)
    # the cheapest linked pair, a middling one and the cheapest patch against itself
    picks = [filled[order[0]], filled[order[len(order) // 3]]]
    diag = [pq for pq in filled[order] if pq[0] == pq[1]]
    picks.append(diag[0])
    jobs = np.array(picks, dtype=np.int32)
    assert cost[order[len(order) // 3]] < 2.5e10  # a few seconds of brute force
    _, sums = oracle.count_jobs(_as_cat(lref), _as_cat(lunk), jobs, t)
    per_scale = combine(np.moveaxis(sums, 0, -1))  # [S, B, J]
    for j, (p, q) in enumerate(jobs):
        np.testing.assert_allclose(per_scale[:, :, j], exp[:, :, p, q], rtol=1e-10, atol=0)
        assert per_scale[:, :, j].sum() > 0
