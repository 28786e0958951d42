"""A strongly clustered synthetic survey (used by tools/make_golden_clustered.py in the build container to run the
REFERENCE on it, and by tests/test_gpu_clustered.py to run the GPU path on the same columns).

Uniform test data keeps every band of the band kernel near its mean length. Here 70 % of the objects sit in 300
Gaussian clumps of 0.02-0.8 degrees inside a 25 degree cap: per-strip densities vary by orders of magnitude, windows
span several LDS stages next to windows with a handful of entries, patches differ tenfold in size, and the redshift of
a reference object follows its clump (the bins see different clustering)."""
import numpy as np

CAP_DEG = 25.0
N_CLUMPS = 300
N_PATCHES = 24
N_BINS = 12
Z_RANGE = (0.1, 1.3)
SCALES_ARCMIN = ([0.3, 3.0], [3.0, 30.0])  # two scales sharing an edge: E = 3 distinct edges per bin


def _cap_points(rng, n, half_opening_deg):
    """Uniform on a spherical cap around the +x axis -> unit vectors."""
    cos_t = rng.uniform(np.cos(np.deg2rad(half_opening_deg)), 1.0, n)
    sin_t = np.sqrt(1.0 - cos_t**2)
    phi = rng.uniform(0.0, 2.0 * np.pi, n)
    return np.column_stack([cos_t, sin_t * np.cos(phi), sin_t * np.sin(phi)])


def _to_radec(v):
    v = v / np.linalg.norm(v, axis=1)[:, None]
    ra = np.arctan2(v[:, 1], v[:, 0]) % (2.0 * np.pi)
    dec = np.arcsin(np.clip(v[:, 2], -1.0, 1.0))
    return ra, dec


def clumps(seed=11):
    rng = np.random.default_rng(seed)
    centres = _cap_points(rng, N_CLUMPS, CAP_DEG - 2.0)
    sigma = np.deg2rad(10.0 ** rng.uniform(np.log10(0.02), np.log10(0.8), N_CLUMPS))
    share = rng.dirichlet(np.full(N_CLUMPS, 0.6))
    z_of = rng.uniform(Z_RANGE[0] + 0.05, Z_RANGE[1] - 0.05, N_CLUMPS)
    return centres, sigma, share, z_of


def sample(seed, n, clustered_fraction=0.7, with_z=True, with_w=False):
    """dict(ra, dec[, z][, w]) in radian; deterministic in (seed, n)."""
    rng = np.random.default_rng(seed)
    centres, sigma, share, z_of = clumps()
    n_cl = int(n * clustered_fraction)
    which = rng.choice(N_CLUMPS, size=n_cl, p=share)
    v_cl = centres[which] + sigma[which, None] * rng.normal(size=(n_cl, 3))
    v_un = _cap_points(rng, n - n_cl, CAP_DEG)
    ra, dec = _to_radec(np.concatenate([v_cl, v_un]))
    out = dict(ra=ra, dec=dec)
    if with_z:
        z = np.concatenate([z_of[which] + 0.03 * rng.normal(size=n_cl), rng.uniform(*Z_RANGE, n - n_cl)])
        out["z"] = z  # a few per cent fall outside the binning and are dropped
    if with_w:
        out["w"] = rng.uniform(0.5, 1.5, n)
    perm = rng.permutation(n)  # no order in the input
    return {k: v[perm] for k, v in out.items()}


def patch_centers(seed=5):
    """(ra, dec) of N_PATCHES centres: half of them on the biggest clumps (small dense patches), half anywhere in the cap."""
    rng = np.random.default_rng(seed)
    centres, _, share, _ = clumps()
    top = centres[np.argsort(share)[::-1][: N_PATCHES // 2]]
    rest = _cap_points(rng, N_PATCHES - len(top), CAP_DEG)
    ra, dec = _to_radec(np.concatenate([top, rest]))
    return np.column_stack([ra, dec])


def bin_edges():
    return np.linspace(Z_RANGE[0] + 0.1, Z_RANGE[1] - 0.1, N_BINS + 1)
