"""GPU: catalogue preparation on the device -- patch assignment (``yawhip_assign_patches``, replaces
scipy.cluster.vq.vq of assign_patch_centers, catalog.py:229-249) and the ordering of a catalogue at
upload (rocPRIM sorts in yawhip_sort.hip). Patch ids must be identical to scipy's including exact
ties; the upload order is library-private, so it is checked through what it must guarantee."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sphere(rng, n):
    ra = rng.uniform(0, 2 * np.pi, n)
    dec = np.arcsin(rng.uniform(-1, 1, n))
    return np.column_stack([np.cos(ra) * np.cos(dec), np.sin(ra) * np.cos(dec), np.sin(dec)])


def test_assign_patches_matches_scipy_vq():
    from scipy.cluster.vq import vq

    from yet_another_wizz_amd import engine

    rng = np.random.default_rng(5)
    xyz = _sphere(rng, 1_000_000)
    centers = _sphere(rng, 64)
    # exact ties: centres mirrored at the plane x = 0 and objects on that plane -> first minimum wins
    centers[3] = np.array([0.05, 0.6, 0.8]) / np.linalg.norm([0.05, 0.6, 0.8])
    centers[20] = np.array([0.05, -0.7, 0.1]) / np.linalg.norm([0.05, -0.7, 0.1])
    centers[10] = centers[3] * [-1, 1, 1]
    centers[40] = centers[20] * [-1, 1, 1]
    on_plane = _sphere(rng, 5000)
    on_plane[:, 0] = 0.0
    on_plane /= np.linalg.norm(on_plane, axis=1)[:, None]
    xyz = np.concatenate([xyz, on_plane])
    expect, _ = vq(xyz, centers)
    got = engine.assign_patches(xyz, centers)
    assert got is not None and got.dtype == np.int64
    assert np.array_equal(got, expect)
    tied = got[-5000:]
    assert np.isin(tied, [3, 20]).sum() > 50 and not np.isin(tied, [10, 40]).any()


def test_catalog_from_arrays_uses_the_same_ids(monkeypatch):
    import yet_another_wizz_amd as yaw
    from yet_another_wizz_amd import catalog, engine

    rng = np.random.default_rng(6)
    n = 400_000
    ra, dec = rng.uniform(0, 2 * np.pi, n), np.arcsin(rng.uniform(-1, 1, n))
    cxyz = _sphere(rng, 24)
    centers = yaw.AngularCoordinates(np.column_stack([np.arctan2(cxyz[:, 1], cxyz[:, 0]) % (2 * np.pi), np.arcsin(cxyz[:, 2])]))
    calls = []
    orig = engine.assign_patches
    monkeypatch.setattr(engine, "assign_patches", lambda *a: calls.append(1) or orig(*a))
    dev = yaw.Catalog.from_arrays(ra, dec, patch_centers=centers, degrees=False)
    assert calls  # the device did the assignment
    monkeypatch.setattr(catalog, "DEVICE_ASSIGN_MIN", 10**12)  # host path (scipy)
    host = yaw.Catalog.from_arrays(ra, dec, patch_centers=centers, degrees=False)
    assert np.array_equal(dev.get_num_records(), host.get_num_records())
    assert np.array_equal(dev.get_centers().data, host.get_centers().data)


def test_upload_orders_segments_and_runs():
    """What the device-side ordering must guarantee, seen through the ABI: counts do not depend on the order
    of the objects inside a (patch, bin) segment of the input."""
    from oracle import oracle
    from yet_another_wizz_amd import _lib

    rng = np.random.default_rng(7)
    n, P, B = 60000, 3, 4
    ra = np.deg2rad(rng.uniform(10, 16, n)); dec = np.deg2rad(rng.uniform(-3, 3, n))
    patch = rng.integers(0, P, n); z = rng.uniform(0.1, 0.9, n)
    cat = oracle.sort_catalog(ra, dec, z, None, patch, P, np.linspace(0.1, 0.9, B + 1), "right")
    unk = oracle.sort_catalog(ra[::2] + 1e-3, dec[::2], z[::2], None, patch[::2], P, None, "right")
    shuffled = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in cat.items()}
    for s in range(P * B):  # another order inside every segment
        lo, hi = cat["off"][s], cat["off"][s + 1]
        perm = rng.permutation(hi - lo) + lo
        for col in ("x", "y", "z"):
            shuffled[col][lo:hi] = cat[col][perm]
    ctx = _lib.Context(0)
    jobs = np.array([(p, q) for p in range(P) for q in range(P)], dtype=np.int32)
    t = np.tile(oracle.thresholds_for(oracle.ang_bins_for(oracle.parse_ang_limits([1 * np.pi / 10800], [9 * np.pi / 10800]), None, None)), (B, 1))
    up = lambda c, nb: _lib.DeviceCatalog(ctx, c["x"], c["y"], c["z"], None, P, nb, c["off"])
    a, _, _ = _lib.count_pairs(ctx, up(cat, B), up(unk, 1), jobs, t)
    b, _, _ = _lib.count_pairs(ctx, up(shuffled, B), up(unk, 1), jobs, t)
    exp, _ = oracle.count_jobs(cat, unk, jobs, t)
    assert np.array_equal(a, exp) and np.array_equal(b, exp) and exp.sum() > 1000
    ctx.close()
