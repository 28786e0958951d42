#!/usr/bin/env python3
"""Generate the golden input/output vectors under ``tests/golden/`` by RUNNING THE REFERENCE.

Runs only in the build container (needs ``/root/reference``); the produced ``.npz`` files are
pure data (inputs + the reference's outputs) and are what travels to the GPU box.

Vector groups:
  G1  greatcircle.npz   the closed-form point set of tests/catalog/test_trees.py:134-159 and the
                        outputs of ``AngularTree.count`` for the cases of :181-247
  G2-G5 single_job.npz  one patch pair, many (scales, rweight, weights, closed, auto) variants,
                        captured at the ``AngularTree.count`` seam (trees.py:303-362) including
                        the *fine* counts scipy returned (trees.py:348-356)
  G6  full_cross.npz / full_auto.npz   8-patch ``yaw.crosscorrelate`` (DD, DR, RD, RR) and
                        ``yaw.autocorrelate`` (DD, DR, RR) tensors + ``CorrFunc.sample()``
  G7  twodflens.npz     bundled 2dFLenS example catalogue (11 patches, weights) in arcmin units

Usage:  python tools/make_golden.py
"""
from __future__ import annotations

import math
import os
import shutil
import sys
import tempfile
from itertools import product

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_loader import load_reference  # noqa: E402

yaw = load_reference()
import pandas as pd  # noqa: E402
from yaw.catalog import trees as rtrees  # noqa: E402
from yaw.coordinates import AngularCoordinates, AngularDistances  # noqa: E402
from yaw.correlation.measurements import PatchLinkage  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
ARCMIN = math.pi / 180.0 / 60.0


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# --------------------------------------------------------------------------- G1
def great_circle_points_deg():
    """Point set of reference tests/catalog/test_trees.py:134-159 (degrees)."""
    points = np.array(
        [[0.0, 0.0], [90.0, 0.0], [180.0, 0.0], [270.0, 0.0], [0.0, 90.0], [0.0, -90.0]]
    )
    base = np.arange(1.0, 90.0, 1.0)
    for offset in (0.0, 90.0, 180.0, 270.0):
        points = np.concatenate([points, np.column_stack([base + offset, np.zeros_like(base)])])
    for sign, ra in product([-1.0, 1.0], [0.0, 180.0]):
        points = np.concatenate([points, np.column_stack([np.full_like(base, ra), sign * base])])
    for sign, ra in product([-1.0, 1.0], [90.0, 270.0]):
        points = np.concatenate([points, np.column_stack([np.full_like(base, ra), sign * base])])
    return points


def make_greatcircle():
    DELTA = 1e-9
    pts = AngularCoordinates(np.deg2rad(great_circle_points_deg()))
    w = np.full(len(pts), 2.0)
    tree = rtrees.AngularTree(pts, w)
    single = rtrees.AngularTree(AngularCoordinates([0.0, 0.0]), [2.0])
    out = dict(
        radec=pts.data, xyz=tree.data, single_radec=np.array([[0.0, 0.0]]), single_xyz=single.data
    )
    for am in (1.0, 2.0, 10.0, 89.0):
        hi = am + DELTA
        out[f"single_{int(am)}"] = tree.count(single, np.deg2rad(hi - 1.0), np.deg2rad(hi))
        out[f"range_{int(am)}"] = tree.count(single, DELTA, np.deg2rad(am) + DELTA)
    for am in (2.0, 10.0, 89.0):
        hi = np.arange(1.0, am) + DELTA
        out[f"bins_{int(am)}"] = tree.count(single, np.deg2rad(hi - 1.0), np.deg2rad(hi))
    utree = rtrees.AngularTree(pts)
    lims = np.deg2rad([0.0, 1.0]) + DELTA
    out["dualtree"] = utree.count(utree, *lims)
    save("greatcircle.npz", **out)


# --------------------------------------------------------------------------- G2-G5
def ref_count_with_fine(tree1, tree2, ang_min, ang_max, weight_scale, weight_res):
    """Re-trace AngularTree.count (trees.py:339-362) with the reference's own helpers, keeping
    the intermediate fine-bin counts; asserts the result equals AngularTree.count itself."""
    ang_limits = rtrees.parse_ang_limits(ang_min, ang_max)
    ang_bins = rtrees.get_ang_bins(ang_limits, weight_scale, weight_res)
    cumulative = len(ang_bins) < 8
    r = AngularDistances(ang_bins).to_3d()
    raw = tree1.tree.count_neighbors(
        tree2.tree, r=r, weights=(tree1.weights, tree2.weights), cumulative=cumulative
    ).astype(np.float64)
    fine = rtrees.dispatch_counts(raw, cumulative)
    final = tree1.count(tree2, ang_min, ang_max, weight_scale=weight_scale, weight_res=weight_res)
    chk = fine.copy()
    if weight_scale is not None:
        aw = rtrees.logarithmic_mid(ang_bins) ** weight_scale
        chk *= aw / aw.sum()
    chk = rtrees.get_counts_for_limits(chk, ang_bins, ang_limits)
    assert np.array_equal(chk, final)
    thresholds = np.array([math.pow(x, 2.0) for x in r])
    return ang_bins, thresholds, fine, final


SCALE_SETS = {
    "s1": ([1.0], [10.0]),
    "s2over": ([1.0, 5.0], [10.0, 20.0]),
    "s3log": ([0.5, 1.58, 5.0], [1.58, 5.0, 15.8]),
    "s4gap": ([0.5, 2.0, 4.0, 8.0], [1.0, 3.0, 6.0, 16.0]),
}


def make_single_job():
    rng = np.random.default_rng(12345)
    n1, n2 = 1500, 2000
    box = 1.5  # degrees

    def draw(n):
        ra = rng.uniform(10.0, 10.0 + box, n)
        dec = np.rad2deg(np.arcsin(rng.uniform(np.sin(np.deg2rad(-20.0)), np.sin(np.deg2rad(-20.0 + box)), n)))
        return np.deg2rad(np.column_stack([ra, dec]))

    radec1, radec2 = draw(n1), draw(n2)
    w1, w2 = rng.uniform(0.5, 1.5, n1), rng.uniform(0.5, 1.5, n2)
    c1, c2 = AngularCoordinates(radec1), AngularCoordinates(radec2)
    out = dict(radec1=radec1, radec2=radec2, w1=w1, w2=w2)
    trees = {
        "uu": (rtrees.AngularTree(c1), rtrees.AngularTree(c2)),
        "ww": (rtrees.AngularTree(c1, w1), rtrees.AngularTree(c2, w2)),
        "wu": (rtrees.AngularTree(c1, w1), rtrees.AngularTree(c2)),
        "uw": (rtrees.AngularTree(c1), rtrees.AngularTree(c2, w2)),
    }
    out["xyz1"] = trees["uu"][0].data
    out["xyz2"] = trees["uu"][1].data
    names = []
    for sname, (lo, hi) in SCALE_SETS.items():
        amin, amax = np.array(lo) * ARCMIN, np.array(hi) * ARCMIN
        for wname, (t1, t2) in trees.items():
            for rw, res in ((None, None), (-1.0, 20), (0.5, 7)):
                if rw is not None and wname in ("wu", "uw"):
                    continue
                for kind in ("cross", "auto"):
                    if kind == "auto" and wname in ("wu", "uw"):
                        continue
                    other = t2 if kind == "cross" else t1
                    key = f"{sname}.{wname}.{'plain' if rw is None else f'rw{rw}_{res}'}.{kind}"
                    ang_bins, thr, fine, final = ref_count_with_fine(
                        t1, other, amin, amax, rw, res if res is not None else 50
                    )
                    out[key + ".ang_bins"] = ang_bins
                    out[key + ".t"] = thr
                    out[key + ".fine"] = fine
                    out[key + ".final"] = final
                    names.append(key)
    out["case_names"] = np.array(names)
    for sname, (lo, hi) in SCALE_SETS.items():
        out[f"scales.{sname}"] = np.array([lo, hi])
    save("single_job.npz", **out)


# --------------------------------------------------------------------------- G6
def box_catalog_frame(rng, n, ra0, ra1, dec0, dec1, *, redshifts, weights, zgrid=None):
    ra = rng.uniform(ra0, ra1, n)
    dec = np.rad2deg(np.arcsin(rng.uniform(np.sin(np.deg2rad(dec0)), np.sin(np.deg2rad(dec1)), n)))
    cols = dict(ra=ra, dec=dec)
    if redshifts:
        z = rng.uniform(0.05, 1.05, n)  # some objects fall outside [0.1, 1.0]
        if zgrid is not None:  # put a share of objects exactly on bin edges (closed-side test)
            on_edge = rng.random(n) < 0.15
            z[on_edge] = rng.choice(zgrid, on_edge.sum())
        cols["z"] = z
    if weights:
        cols["w"] = rng.uniform(0.5, 1.5, n)
    return pd.DataFrame(cols)


def grid_centers(ra0, ra1, dec0, dec1, nra, ndec):
    ras = ra0 + (np.arange(nra) + 0.5) * (ra1 - ra0) / nra
    decs = dec0 + (np.arange(ndec) + 0.5) * (dec1 - dec0) / ndec
    rr, dd = np.meshgrid(ras, decs)
    return np.deg2rad(np.column_stack([rr.ravel(), dd.ravel()]))


def dump_counts(prefix, corrfuncs, out):
    for s, cf in enumerate(corrfuncs):
        for kind in ("dd", "dr", "rd", "rr"):
            nc = getattr(cf, kind)
            if nc is None:
                continue
            out[f"{prefix}.s{s}.{kind}.counts"] = nc.counts.counts
            out[f"{prefix}.s{s}.{kind}.sum_weights1"] = nc.sum_weights.sum_weights1
            out[f"{prefix}.s{s}.{kind}.sum_weights2"] = nc.sum_weights.sum_weights2
            sp = nc.sample_patch_sum()
            out[f"{prefix}.s{s}.{kind}.sample_data"] = sp.data
            out[f"{prefix}.s{s}.{kind}.sample_samples"] = sp.samples
        cd = cf.sample()
        out[f"{prefix}.s{s}.corr_data"] = cd.data
        out[f"{prefix}.s{s}.corr_samples"] = cd.samples
        out[f"{prefix}.s{s}.corr_error"] = cd.error
        out[f"{prefix}.s{s}.corr_covariance"] = cd.covariance


def frame_arrays(prefix, df, out):
    for col in df.columns:
        out[f"{prefix}.{col}"] = df[col].to_numpy()


def catalog_meta(prefix, cat, out):
    out[f"{prefix}.centers"] = cat.get_centers().data
    out[f"{prefix}.radii"] = cat.get_radii().data
    out[f"{prefix}.num_records"] = np.array(cat.get_num_records())
    out[f"{prefix}.sum_weights"] = np.array(cat.get_sum_weights())


def make_full(tmp):
    rng = np.random.default_rng(777)
    ra0, ra1, dec0, dec1 = 30.0, 50.0, -10.0, 0.0
    centers = grid_centers(ra0, ra1, dec0, dec1, 4, 2)
    zedges = np.linspace(0.1, 1.0, 7)
    kw = dict(zgrid=zedges)
    for weighted in (False, True):
        tag = "w" if weighted else "u"
        frames = dict(
            ref=box_catalog_frame(rng, 4000, ra0, ra1, dec0, dec1, redshifts=True, weights=weighted, **kw),
            unk=box_catalog_frame(rng, 5000, ra0, ra1, dec0, dec1, redshifts=False, weights=weighted),
            ref_rand=box_catalog_frame(rng, 9000, ra0, ra1, dec0, dec1, redshifts=True, weights=False, **kw),
            unk_rand=box_catalog_frame(rng, 11000, ra0, ra1, dec0, dec1, redshifts=False, weights=weighted),
        )
        cats = {}
        for name, df in frames.items():
            cats[name] = yaw.Catalog.from_dataframe(
                os.path.join(tmp, f"full_{tag}_{name}"),
                df,
                ra_name="ra",
                dec_name="dec",
                weight_name="w" if "w" in df.columns else None,
                redshift_name="z" if "z" in df.columns else None,
                patch_centers=AngularCoordinates(centers),
                overwrite=True,
            )
        inputs = dict(patch_centers=centers, zedges=zedges)
        for name, df in frames.items():
            frame_arrays(name, df, inputs)
            catalog_meta(name + ".meta", cats[name], inputs)
        save(f"full_{tag}_inputs.npz", **inputs)
        for closed in ("right", "left"):
            for cfgname, cfgkw in (
                ("s2", dict(rmin=[2.0, 5.0], rmax=[20.0, 40.0], unit="arcmin")),
                ("rw", dict(rmin=[2.0], rmax=[30.0], unit="arcmin", rweight=-0.8, resolution=12)),
            ):
                if closed == "left" and cfgname == "rw":
                    continue
                config = yaw.Configuration.create(edges=zedges, closed=closed, **cfgkw)
                out = dict(zedges=zedges)
                links = PatchLinkage.from_catalogs(config, cats["ref"], cats["unk"], cats["ref_rand"], cats["unk_rand"])
                out["cross.job_pairs"] = np.array(sorted(links.iter_patch_id_pairs(auto=False)))
                out["auto.job_pairs"] = np.array(sorted(links.iter_patch_id_pairs(auto=True)))
                cfs = yaw.crosscorrelate(
                    config, cats["ref"], cats["unk"], ref_rand=cats["ref_rand"], unk_rand=cats["unk_rand"]
                )
                dump_counts("cross", cfs, out)
                cfs = yaw.autocorrelate(config, cats["ref"], cats["ref_rand"], count_rr=True)
                dump_counts("auto", cfs, out)
                save(f"full_{tag}_{cfgname}_{closed}.npz", **out)


# --------------------------------------------------------------------------- G7
def make_2dflens(tmp):
    data = pd.read_parquet("/root/reference/src/yaw/examples/2dflens_kidss_data.pqt")
    rand = pd.read_parquet("/root/reference/src/yaw/examples/2dflens_kidss_rand_5x.pqt")
    rand = rand.iloc[::5].reset_index(drop=True)  # keep the fixture small: 1x randoms
    print("2dflens columns:", list(data.columns), len(data), len(rand))
    cats = {}
    for name, df in (("data", data), ("rand", rand)):
        cats[name] = yaw.Catalog.from_dataframe(
            os.path.join(tmp, f"flens_{name}"),
            df,
            ra_name="RA",
            dec_name="Dec",
            weight_name="wei",
            redshift_name="redshift",
            patch_name="patch",
            overwrite=True,
        )
    # second, redshift-free view of the same objects as the "unknown" sample
    unk = yaw.Catalog.from_dataframe(
        os.path.join(tmp, "flens_unk"),
        data,
        ra_name="RA",
        dec_name="Dec",
        weight_name="wei",
        patch_name="patch",
        overwrite=True,
    )
    config = yaw.Configuration.create(rmin=1.0, rmax=10.0, unit="arcmin", zmin=0.15, zmax=0.7, num_bins=11)
    out = dict(zedges=np.asarray(config.binning.binning.edges))
    frame_arrays("data", data[["RA", "Dec", "redshift", "wei", "patch"]], out)
    frame_arrays("rand", rand[["RA", "Dec", "redshift", "wei", "patch"]], out)
    catalog_meta("data.meta", cats["data"], out)
    catalog_meta("rand.meta", cats["rand"], out)
    cfs = yaw.crosscorrelate(config, cats["data"], unk, ref_rand=cats["rand"])
    dump_counts("cross", cfs, out)
    cfs = yaw.autocorrelate(config, cats["data"], cats["rand"], count_rr=True)
    dump_counts("auto", cfs, out)
    save("twodflens.npz", **out)


def make_refcache(tmp):
    """A small catalogue cache written by the reference (data.bin / meta.yml / patch_ids.bin): the
    on-disk input contract of the path (patch.py:164-178, catalog.py:325-331)."""
    rng = np.random.default_rng(4242)
    df = box_catalog_frame(rng, 900, 100.0, 104.0, 20.0, 23.0, redshifts=True, weights=True)
    centers = grid_centers(100.0, 104.0, 20.0, 23.0, 2, 2)
    path = os.path.join(tmp, "refcache")
    cat = yaw.Catalog.from_dataframe(path, df, ra_name="ra", dec_name="dec", weight_name="w", redshift_name="z",
                                     patch_centers=AngularCoordinates(centers), overwrite=True)
    dest = os.path.join(OUT, "refcache")
    shutil.rmtree(dest, ignore_errors=True)
    shutil.copytree(path, dest)
    out = {}
    frame_arrays("input", df, out)
    catalog_meta("meta", cat, out)
    out["patch_centers"] = centers
    save("refcache_expect.npz", **out)
    print("refcache files:", sorted(os.listdir(dest)), sorted(os.listdir(os.path.join(dest, "patch_0"))))


def make_refcache_counts(tmp):
    """The reference's own measurements taken FROM the committed reference-written cache (tests/golden/refcache):
    crosscorrelate (DD, DR) with the cache as the reference sample and autocorrelate (DD, DR, RR) of the cache against
    a second copy of the frame. Only the outputs are new data; the inputs are refcache/ and refcache_expect.npz."""
    exp = np.load(os.path.join(OUT, "refcache_expect.npz"))
    df = pd.DataFrame({c: exp[f"input.{c}"] for c in ("ra", "dec", "z", "w")})
    centers = AngularCoordinates(exp["patch_centers"])
    cache = os.path.join(tmp, "refcache_copy")  # build_trees writes into the cache directory: work on a copy
    shutil.copytree(os.path.join(OUT, "refcache"), cache)
    ref = yaw.Catalog(cache)
    kw = dict(ra_name="ra", dec_name="dec", weight_name="w", patch_centers=centers, overwrite=True)
    unk = yaw.Catalog.from_dataframe(os.path.join(tmp, "rc_unk"), df, **kw)
    rnd = yaw.Catalog.from_dataframe(os.path.join(tmp, "rc_rnd"), df, **kw)
    rnz = yaw.Catalog.from_dataframe(os.path.join(tmp, "rc_rnz"), df, redshift_name="z", **kw)
    config = yaw.Configuration.create(rmin=0.5, rmax=8.0, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=4)
    out = {}
    dump_counts("cross", yaw.crosscorrelate(config, ref, unk, unk_rand=rnd), out)
    dump_counts("auto", yaw.autocorrelate(config, ref, rnz, count_rr=True), out)
    save("refcache_counts.npz", **out)


def main():
    tmp = tempfile.mkdtemp(prefix="yawgolden_", dir="/dev/shm")
    try:
        if "--refcache-counts" in sys.argv:  # only this group (the others are already committed)
            make_refcache_counts(tmp)
            return
        make_refcache(tmp)
        make_greatcircle()
        make_single_job()
        make_full(tmp)
        make_2dflens(tmp)
        make_refcache_counts(tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
