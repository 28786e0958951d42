"""Pin the CPU oracle (oracle/) against vectors captured from the reference itself
(tools/make_golden.py) and against the closed-form known answers of the reference's own
tests/catalog/test_trees.py:181-247.  CPU only."""
import numpy as np
import pytest

from conftest import ARCMIN, load_golden
from oracle import oracle

RTOL_W = 1e-10  # weighted sums: scipy's summation order is tree dependent (north_star tolerance)


def xyz_cols(a):
    return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()


# ------------------------------------------------------------------ G1 great circles
@pytest.fixture(scope="module")
def gc():
    return load_golden("greatcircle.npz")


DELTA = 1e-9


def test_to_3d_matches_reference(gc):
    x, y, z = oracle.to_3d(gc["radec"][:, 0], gc["radec"][:, 1])
    assert np.array_equal(np.column_stack([x, y, z]), gc["xyz"])


@pytest.mark.parametrize("ang_max", [1.0, 2.0, 10.0, 89.0])
def test_greatcircle_single(gc, ang_max):
    # reference test_trees.py:181-195: one 1-degree annulus holds 4 points, weights 2*2
    w = np.full(len(gc["xyz"]), 2.0)
    hi = ang_max + DELTA
    got = oracle.angular_tree_count(
        xyz_cols(gc["xyz"]), w, xyz_cols(gc["single_xyz"]), np.array([2.0]), np.deg2rad(hi - 1.0), np.deg2rad(hi)
    )
    assert got == 4 * 2.0**2
    assert np.array_equal(got, gc[f"single_{int(ang_max)}"])


@pytest.mark.parametrize("ang_max", [2.0, 10.0, 89.0])
def test_greatcircle_bins(gc, ang_max):
    # reference test_trees.py:197-212 (89 annuli -> E >= 8 -> non-cumulative branch)
    w = np.full(len(gc["xyz"]), 2.0)
    hi = np.arange(1.0, ang_max) + DELTA
    got = oracle.angular_tree_count(
        xyz_cols(gc["xyz"]), w, xyz_cols(gc["single_xyz"]), np.array([2.0]), np.deg2rad(hi - 1.0), np.deg2rad(hi)
    )
    assert np.array_equal(got, np.full_like(hi, 16.0))
    assert np.array_equal(got, gc[f"bins_{int(ang_max)}"])


@pytest.mark.parametrize("ang_max", [1.0, 2.0, 10.0, 89.0])
def test_greatcircle_range(gc, ang_max):
    # reference test_trees.py:214-225
    w = np.full(len(gc["xyz"]), 2.0)
    got = oracle.angular_tree_count(
        xyz_cols(gc["xyz"]), w, xyz_cols(gc["single_xyz"]), np.array([2.0]), DELTA, np.deg2rad(ang_max) + DELTA
    )
    assert got == int(ang_max) * 4 * 4.0
    assert np.array_equal(got, gc[f"range_{int(ang_max)}"])


@pytest.mark.parametrize("num_bins", [1, 2])
def test_empty(num_bins):
    # reference test_trees.py:227-237
    e = (np.empty(0), np.empty(0), np.empty(0))
    lo = np.linspace(0.0, 1.0, num_bins) + DELTA
    got = oracle.angular_tree_count(e, np.empty(0), e, np.empty(0), lo, lo + 1.0)
    assert np.array_equal(got, np.zeros(num_bins))


def test_greatcircle_dualtree(gc):
    # reference test_trees.py:239-247: ordered pairs, self pairs excluded
    cols = xyz_cols(gc["xyz"])
    lims = np.deg2rad([0.0, 1.0]) + DELTA
    got = oracle.angular_tree_count(cols, None, cols, None, *lims)
    assert got == 4 * 6 + 2 * (len(gc["xyz"]) - 6)
    assert np.array_equal(got, gc["dualtree"])


def test_invalid_angles(gc):
    # reference test_trees.py:249-254
    cols = xyz_cols(gc["xyz"])
    with pytest.raises(ValueError):
        oracle.angular_tree_count(cols, None, cols, None, [-1.0], [1.0])
    with pytest.raises(ValueError):
        oracle.angular_tree_count(cols, None, cols, None, [1.0], [np.pi + DELTA])


# ------------------------------------------------------------------ host helpers (test_trees.py:56-131)
@pytest.mark.parametrize(
    "rng,expect",
    [
        ([[1.0, 10.0], [10.0, 100.0]], [1.0, 10.0, 100.0]),
        ([[1.0, 10.0], [11.0, 100.0]], [1.0, 10.0, 11.0, 100.0]),
        ([[1.0, 11.0], [10.0, 100.0]], [1.0, 10.0, 11.0, 100.0]),
    ],
)
def test_ang_bins(rng, expect):
    np.testing.assert_almost_equal(oracle.ang_bins_for(np.array(rng), None, None), expect)


@pytest.mark.parametrize(
    "rng,expect",
    [
        ([[0.1, 9.0], [9.0, 1000.0]], [0.1, 1.0, 9.0, 10.0, 100.0, 1000.0]),
        ([[0.1, 10.0], [10.0, 1000.0]], [0.1, 1.0, 10.0, 100.0, 1000.0]),
        ([[0.1, 10.0], [11.0, 1000.0]], [0.1, 1.0, 10.0, 11.0, 100.0, 1000.0]),
        ([[0.1, 11.0], [10.0, 1000.0]], [0.1, 1.0, 10.0, 11.0, 100.0, 1000.0]),
    ],
)
def test_ang_bins_weights(rng, expect):
    np.testing.assert_almost_equal(oracle.ang_bins_for(np.array(rng), 1.0, 4), expect)


@pytest.mark.parametrize(
    "lims,expect",
    [([[1.0, 1000.0]], [3]), ([[10.0, 1000.0]], [2]), ([[1.0, 10.0], [100.0, 1000.0]], [1, 1]),
     ([[1.0, 10.0], [10.0, 1000.0]], [1, 2])],
)
def test_counts_for_limits(lims, expect):
    ang_bins = np.array([1.0, 10.0, 100.0, 1000.0])
    got = oracle.finalize(np.ones(3), ang_bins, np.array(lims), None)
    assert np.array_equal(got, expect)


# ------------------------------------------------------------------ G2-G5 single job
@pytest.fixture(scope="module")
def sj():
    return load_golden("single_job.npz")


def _case_inputs(sj, key):
    sname, wname, rwname, kind = key.split(".", 3) if key.count(".") == 3 else _split(key)
    lo, hi = sj[f"scales.{sname}"]
    w1 = sj["w1"] if wname[0] == "w" else None
    w2 = sj["w2"] if wname[1] == "w" else None
    a = xyz_cols(sj["xyz1"])
    if kind == "auto":
        b, w2 = a, w1
    else:
        b = xyz_cols(sj["xyz2"])
    if rwname == "plain":
        rw, res = None, 50
    else:
        rw, res = rwname[2:].rsplit("_", 1)
        rw, res = float(rw), int(res)
    return a, w1, b, w2, lo * ARCMIN, hi * ARCMIN, rw, res


def _split(key):
    # keys look like "s1.ww.rw-1.0_20.cross": the rweight token itself contains a dot
    parts = key.split(".")
    return parts[0], parts[1], ".".join(parts[2:-1]), parts[-1]


def test_single_job_all_cases(sj):
    names = [str(n) for n in sj["case_names"]]
    assert len(names) >= 40
    for key in names:
        a, w1, b, w2, lo, hi, rw, res = _case_inputs(sj, key)
        lim = oracle.parse_ang_limits(lo, hi)
        ang_bins = oracle.ang_bins_for(lim, rw, res)
        assert np.array_equal(ang_bins, sj[key + ".ang_bins"]), key
        t = oracle.thresholds_for(ang_bins)
        assert np.array_equal(t, sj[key + ".t"]), key
        counts, sums = oracle.count_tree(a, w1, b, w2, t)
        final = oracle.angular_tree_count(a, w1, b, w2, lo, hi, rw, res)
        if w1 is None and w2 is None:
            assert np.array_equal(counts.astype(np.float64), sj[key + ".fine"]), key
            assert np.array_equal(final, sj[key + ".final"]), key  # bit-identical
        else:
            np.testing.assert_allclose(sums, sj[key + ".fine"], rtol=RTOL_W, atol=0, err_msg=key)
            np.testing.assert_allclose(final, sj[key + ".final"], rtol=RTOL_W, atol=0, err_msg=key)


def test_numpy_restatement_equals_c(sj):
    a, b = xyz_cols(sj["xyz1"][:400]), xyz_cols(sj["xyz2"][:500])
    t = sj["s4gap.uu.plain.cross.t"]
    c1, s1 = oracle.count_tree(a, sj["w1"][:400], b, sj["w2"][:500], t)
    c2, s2 = oracle.count_tree_numpy(a, sj["w1"][:400], b, sj["w2"][:500], t)
    assert np.array_equal(c1, c2)
    np.testing.assert_allclose(s1, s2, rtol=1e-13)


# ------------------------------------------------------------------ G6 full crosscorrelate / autocorrelate
def _oracle_cat(inp, name, n_patches, edges, closed, centers_xyz):
    ra, dec = np.deg2rad(inp[f"{name}.ra"]), np.deg2rad(inp[f"{name}.dec"])
    x, y, z = oracle.to_3d(ra, dec)
    # nearest centre in xyz (catalog.py:229-249 assign_patch_centers via scipy.cluster.vq)
    d2 = ((np.stack([x, y, z], 1)[:, None, :] - centers_xyz[None]) ** 2).sum(-1)
    patch = d2.argmin(1)
    zz = inp[f"{name}.z"] if f"{name}.z" in inp else None
    w = inp[f"{name}.w"] if f"{name}.w" in inp else None
    return oracle.sort_catalog(ra, dec, zz, w, patch, n_patches, edges, closed)


@pytest.mark.parametrize("tag", ["u", "w"])
@pytest.mark.parametrize("cfg,closed", [("s2", "right"), ("s2", "left"), ("rw", "right")])
def test_full_pipeline(tag, cfg, closed):
    inp = load_golden(f"full_{tag}_inputs.npz")
    exp = load_golden(f"full_{tag}_{cfg}_{closed}.npz")
    edges = inp["zedges"]
    centers = inp["patch_centers"]
    cx, cy, cz = oracle.to_3d(centers[:, 0], centers[:, 1])
    cxyz = np.stack([cx, cy, cz], 1)
    P, B = len(centers), len(edges) - 1
    if cfg == "s2":
        lo, hi, rw, res = np.array([2.0, 5.0]), np.array([20.0, 40.0]), None, 50
    else:
        lo, hi, rw, res = np.array([2.0]), np.array([30.0]), -0.8, 12
    amin = np.tile(lo * ARCMIN, (B, 1))
    amax = np.tile(hi * ARCMIN, (B, 1))
    ref = _oracle_cat(inp, "ref", P, edges, closed, cxyz)
    rnd = _oracle_cat(inp, "ref_rand", P, edges, closed, cxyz)
    unk = _oracle_cat(inp, "unk", P, None, closed, cxyz)
    urd = _oracle_cat(inp, "unk_rand", P, None, closed, cxyz)
    weighted = tag == "w"

    def check(prefix, kind, c1, c2, jobs, auto):
        counts, sw1, sw2 = oracle.count_pairs(c1, c2, jobs, amin, amax, P, auto=auto, rweight=rw, resolution=res)
        for s in range(len(lo)):
            e = exp[f"{prefix}.s{s}.{kind}.counts"]
            if c1["w"] is None and c2["w"] is None and rw is None:
                assert np.array_equal(counts[s], e), (prefix, kind, s)
            else:
                np.testing.assert_allclose(counts[s], e, rtol=RTOL_W, atol=0, err_msg=f"{prefix}.{kind}.{s}")
            np.testing.assert_allclose(sw1, exp[f"{prefix}.s{s}.{kind}.sum_weights1"], rtol=1e-13)
            np.testing.assert_allclose(sw2, exp[f"{prefix}.s{s}.{kind}.sum_weights2"], rtol=1e-13)

    cj, aj = exp["cross.job_pairs"], exp["auto.job_pairs"]
    check("cross", "dd", ref, unk, cj, False)
    check("cross", "dr", ref, urd, cj, False)
    check("cross", "rd", rnd, unk, cj, False)
    check("cross", "rr", rnd, urd, cj, False)
    check("auto", "dd", ref, ref, aj, True)
    # autocorrelate's DR is count_pairs(data, random): two catalogues -> auto=False, all ordered
    # linked pairs, no halving (measurements.py:330,521)
    check("auto", "dr", ref, rnd, cj, False)
    check("auto", "rr", rnd, rnd, aj, True)
    assert weighted == (ref["w"] is not None)
