"""In-memory catalogue split into spatial patches.

Keeps the ``yaw.Catalog`` call shapes (src/yaw/catalog/catalog.py:911-1468): a ``Mapping`` from
patch id (0..P-1) to :class:`Patch`, the ``from_dataframe`` / ``from_file`` keyword names, and the
per-patch accessors ``get_centers`` / ``get_radii`` / ``get_num_records`` / ``get_sum_weights`` the
measurement driver reads.  What is different by design: the reference caches every patch on disk
(``data.bin`` AoS rows, catalog.py:325-331, patch.py:164-178) and re-reads pickled KD-trees per job;
here a catalogue is a set of float64 columns held once, sorted by patch, and ``build_trees``
prepares the (patch, redshift-bin)-sorted SoA layout that is uploaded to HBM once per measurement.
"""
from __future__ import annotations

from collections.abc import Mapping
from pathlib import Path

import numpy as np

from . import _threads
from .binning import Binning
from .coordinates import AngularCoordinates, AngularDistances
from .options import Closed

radec_to_xyz = _threads.radec_to_xyz  # numpy's cos / sin on slices of the columns, side by side (same values)

__all__ = ["Catalog", "Patch", "Metadata", "InconsistentPatchesError", "PatchLayout"]

PATCH_ID_MAX = np.iinfo(np.int16).max  # the reference stores patch ids as 16-bit integers (datachunk.py:40-42)


class InconsistentPatchesError(Exception):
    """Patches of two catalogues do not line up (catalog.py:80-82)."""


def _check_patch_count(num: int) -> None:
    if num < 1 or num > PATCH_ID_MAX:
        raise ValueError(f"number of patches must be in range [1, {PATCH_ID_MAX}]")


def _stable_argsort_small(keys, num_values: int):
    """Stable argsort of non-negative integer keys < ``num_values``. numpy sorts 16-bit integers with a
    radix sort (an order of magnitude faster than the merge sort it uses for int64); the permutation is
    the same, stable sorts being unique."""
    if num_values <= np.iinfo(np.uint16).max:
        return np.argsort(keys.astype(np.uint16), kind="stable")
    return np.argsort(keys, kind="stable")


DEVICE_ASSIGN_MIN = 200_000  # below this the host is as fast as a round trip to the device
HOST_GROUP_MIN = 200_000     # below this numpy's argsort + gathers are as fast as the library's threaded counting sort


def nearest_center(xyz, centers_xyz, chunk: int = 1 << 18):
    """Index of the nearest centre in Euclidean xyz for every object; ``xyz`` is an [N, 3] array or a tuple of
    three columns.

    Same rule as ``assign_patch_centers`` (catalog.py:229-249, scipy.cluster.vq.vq): squared
    distance accumulated x, y, z in that order, first minimum wins."""
    columns = xyz if isinstance(xyz, tuple) else None
    n = len(columns[0]) if columns is not None else len(xyz)
    if n >= DEVICE_ASSIGN_MIN:  # large inputs: on the GPU if there is one (identical ids, 40x faster)
        from . import engine

        ids = engine.assign_patches(columns if columns is not None else xyz, centers_xyz)
        if ids is not None:
            return ids
    if columns is not None:
        xyz = np.column_stack(columns)
    try:
        from scipy.cluster.vq import vq

        ids, _ = vq(xyz, centers_xyz)
        return ids.astype(np.int64)
    except ImportError:  # pragma: no cover - scipy is present in the supported images
        out = np.empty(len(xyz), dtype=np.int64)
        for lo in range(0, len(xyz), chunk):
            blk = xyz[lo : lo + chunk]
            d = (blk[:, None, 0] - centers_xyz[None, :, 0]) ** 2
            d += (blk[:, None, 1] - centers_xyz[None, :, 1]) ** 2
            d += (blk[:, None, 2] - centers_xyz[None, :, 2]) ** 2
            out[lo : lo + chunk] = d.argmin(axis=1)
        return out


def kmeans_centers(xyz, weights, num: int, *, seed: int = 12345, iterations: int = 25) -> AngularCoordinates:
    """Patch centres from spherical k-means on a probe sample (stands in for the treecorr k-means of
    ``create_patch_centers``, catalog.py:183-226; deterministic for a given input)."""
    rng = np.random.default_rng(seed)
    n = len(xyz)
    first = int(rng.integers(n))
    centers = [xyz[first]]
    d2 = ((xyz - centers[0]) ** 2).sum(axis=1)
    for _ in range(1, num):  # k-means++ seeding
        prob = d2 / d2.sum()
        centers.append(xyz[int(rng.choice(n, p=prob))])
        d2 = np.minimum(d2, ((xyz - centers[-1]) ** 2).sum(axis=1))
    centers = np.array(centers)
    w = np.ones(n) if weights is None else np.asarray(weights, dtype=np.float64)
    for _ in range(iterations):
        ids = nearest_center(xyz, centers)
        sums = np.zeros_like(centers)
        for axis in range(3):
            sums[:, axis] = np.bincount(ids, weights=w * xyz[:, axis], minlength=num)
        norm = np.linalg.norm(sums, axis=1)
        moved = np.where(norm[:, None] > 0, sums / np.maximum(norm, 1e-300)[:, None], centers)
        if np.allclose(moved, centers, atol=1e-12):
            break
        centers = moved
    return AngularCoordinates.from_3d(centers)


# ---------------------------------------------------------------------------------------------
# On-disk cache format of the reference (read and written unchanged so that caches are
# interchangeable):  <dir>/patch_ids.bin  int16 ids (catalog.py:325-331,529-530),
# <dir>/patch_<id>/data.bin  1 header byte + row-major float64 records (patch.py:164-178,
# datachunk.py:47,74-117), <dir>/patch_<id>/meta.yml (patch.py:149-161).
PATCH_INFO_FILE = "patch_ids.bin"
PATCH_NAME_TEMPLATE = "patch_{:d}"
PATCH_DATA_FILE = "data.bin"
PATCH_META_FILE = "meta.yml"
_ATTR_BITS = (("ra", 0), ("dec", 1), ("weights", 2), ("redshifts", 3), ("patch_ids", 4), ("kappa", 5))


def read_patch_file(path):
    """data.bin -> dict of float64 columns (header byte: bit set = column present)."""
    with open(path, "rb") as f:
        flags = int.from_bytes(f.read(1), byteorder="big")
        fields = [name for name, bit in _ATTR_BITS if flags & (1 << bit)]
        raw = np.fromfile(f, dtype=np.float64)
    if "ra" not in fields or "dec" not in fields or len(raw) % len(fields):
        raise ValueError(f"not a patch data file: {path}")
    table = raw.reshape(-1, len(fields))
    return {name: table[:, i].copy() for i, name in enumerate(fields)}


def write_patch_file(path, ra, dec, weights=None, redshifts=None) -> None:
    cols = [ra, dec] + [c for c in (weights, redshifts) if c is not None]
    flags = 0b11 | (int(weights is not None) << 2) | (int(redshifts is not None) << 3)
    with open(path, "wb") as f:
        f.write(flags.to_bytes(1, byteorder="big"))
        np.column_stack(cols).astype(np.float64).tofile(f)


class Metadata:
    """Patch summary used to link patches (mirror of patch.py:44-161)."""

    __slots__ = ("num_records", "sum_weights", "center", "radius")

    def __init__(self, *, num_records: int, sum_weights: float, center: AngularCoordinates,
                 radius: AngularDistances) -> None:
        self.num_records = num_records
        self.sum_weights = sum_weights
        self.center = center
        self.radius = radius

    def __repr__(self) -> str:
        return (f"Metadata(num_records={self.num_records}, sum_weights={self.sum_weights}, "
                f"center={self.center.data[0]}, radius={self.radius.data[0]})")

    @classmethod
    def compute(cls, coords: AngularCoordinates, *, weights=None, center: AngularCoordinates | None = None, xyz=None):
        """Sum of weights (or N), weighted mean direction, radius = largest separation from the
        centre (patch.py:103-147). ``xyz`` = ``coords.to_3d()`` if the caller has it already (the
        catalogue computes the unit vectors once): the same values go through the same operations as
        ``coords.mean`` / ``coords.distance``."""
        if center is not None and len(center) != 1:
            raise ValueError("'center' must be one single coordinate")
        sum_weights = float(len(coords)) if weights is None else float(np.sum(weights))
        if xyz is None:
            xyz = coords.to_3d()
        if isinstance(xyz, tuple):  # three columns: the same numbers without an [N, 3] copy
            x, y, z = xyz
            if center is not None:
                centre = center.copy()
            else:  # the reference averages the rows of the [N, 3] array: same call, same summation order
                centre = AngularCoordinates.from_3d(np.average(np.column_stack(xyz), weights=weights, axis=0))
            cx, cy, cz = centre.to_3d()[0]
            chord = np.sqrt(((x - cx) ** 2 + (y - cy) ** 2) + (z - cz) ** 2)  # = ((xyz - c) ** 2).sum(axis=1): left to right
        else:
            centre = center.copy() if center is not None else AngularCoordinates.from_3d(np.average(xyz, weights=weights, axis=0))
            chord = np.sqrt(((xyz - centre.to_3d()) ** 2).sum(axis=1))
        radius = AngularDistances.from_3d(chord).max()
        return cls(num_records=len(coords), sum_weights=sum_weights, center=centre, radius=radius)

    def to_dict(self) -> dict:
        return dict(num_records=int(self.num_records), sum_weights=float(self.sum_weights),
                    center=self.center.tolist()[0], radius=self.radius.tolist()[0])


class Patch:
    """One spatial patch: a contiguous row range of its catalogue's columns."""

    __slots__ = ("meta", "_cat", "_lo", "_hi")

    def __init__(self, catalog: "Catalog", lo: int, hi: int, meta: Metadata) -> None:
        self._cat, self._lo, self._hi, self.meta = catalog, lo, hi, meta

    def __repr__(self) -> str:
        return (f"Patch(num_records={self.meta.num_records}, weights={self.has_weights}, "
                f"redshifts={self.has_redshifts})")

    def __len__(self) -> int:
        return self._hi - self._lo

    @property
    def has_weights(self) -> bool:
        return self._cat._w is not None

    @property
    def has_redshifts(self) -> bool:
        return self._cat._z is not None

    @property
    def coords(self) -> AngularCoordinates:
        sl = slice(self._lo, self._hi)
        return AngularCoordinates(np.column_stack([self._cat._ra[sl], self._cat._dec[sl]]))

    @property
    def weights(self):
        return None if self._cat._w is None else self._cat._w[self._lo : self._hi]

    @property
    def redshifts(self):
        return None if self._cat._z is None else self._cat._z[self._lo : self._hi]


class PatchLayout:
    """Device-ready layout of one catalogue for one redshift binning: float64 SoA columns sorted
    by (patch, bin), CSR offsets over the P*B segments, and the per-segment sum of weights.

    This is the counterpart of ``build_trees`` (src/yaw/catalog/trees.py:365-429): objects outside
    the binning are dropped (:414), an unbinned catalogue has one segment per patch (:400-404)."""

    __slots__ = ("x", "y", "z", "w", "offsets", "num_patches", "num_bins", "sum_weights", "device", "z_extent")

    def __init__(self, x, y, z, w, offsets, num_patches: int, num_bins: int) -> None:
        self.x, self.y, self.z, self.w = x, y, z, w
        self.offsets = offsets
        self.num_patches, self.num_bins = num_patches, num_bins
        if w is None:
            seg = np.diff(offsets).astype(np.float64)  # sum_weights = float(N) without weights (trees.py:225-227)
        else:  # ndarray.sum() per tree (trees.py:233-234); an empty tree has 0.0 (trees.py:249-258)
            seg = np.array([float(w[lo:hi].sum()) for lo, hi in zip(offsets[:-1], offsets[1:])], dtype=np.float64)
        self.sum_weights = seg.reshape(num_patches, num_bins).T.copy()  # [B_or_1, P]
        self.device = {}  # Context id -> DeviceCatalog
        # extent of every patch along the axis the device sorts by (z): feeds the cost model that
        # balances jobs over GPUs (the z-window culling makes flat patches cheaper than tall ones)
        bounds = offsets[:: num_bins]
        self.z_extent = np.array([np.ptp(z[lo:hi]) if hi > lo else 0.0 for lo, hi in zip(bounds[:-1], bounds[1:])])

    @property
    def weighted(self) -> bool:
        return self.w is not None

    @property
    def num_records(self) -> int:
        return len(self.x)

    def segment_sizes(self):
        """int64[P, B_or_1]."""
        return np.diff(self.offsets).reshape(self.num_patches, self.num_bins)

    def sum_weights_for(self, num_bins: int):
        """f64[B, P]; an unbinned catalogue repeats its single tree for every bin (trees.py:600-601)."""
        if self.num_bins == 1 and num_bins != 1:
            return np.repeat(self.sum_weights, num_bins, axis=0)
        return self.sum_weights


class Catalog(Mapping):
    """Catalogue of points on the sphere, split into P spatial patches with ids 0..P-1."""

    def __init__(self, cache_directory, *, max_workers: int | None = None) -> None:
        """Restore a catalogue from a cache directory written by the reference or by :meth:`to_cache`
        (same call as ``yaw.Catalog(cache_directory)``, catalog.py:966-977)."""
        directory = Path(cache_directory)
        if not directory.exists():
            raise OSError(f"cache directory not found: {directory}")
        info = directory / PATCH_INFO_FILE
        if not info.exists():
            raise InconsistentPatchesError("patch info file not found")
        ids = np.fromfile(info, dtype=np.int16).astype(np.int64)
        if not np.array_equal(np.sort(ids), np.arange(len(ids))):
            raise InconsistentPatchesError("patch IDs must be contiguous and start at 0")
        columns, metas = [], {}
        for pid in range(len(ids)):
            patch_dir = directory / PATCH_NAME_TEMPLATE.format(pid)
            columns.append(read_patch_file(patch_dir / PATCH_DATA_FILE))
            meta_file = patch_dir / PATCH_META_FILE
            if meta_file.exists():
                import yaml

                with meta_file.open() as f:
                    metas[pid] = yaml.safe_load(f)
        present = [tuple(sorted(c)) for c in columns]
        if any(p != present[0] for p in present):
            raise InconsistentPatchesError("data columns are not consistent between patches")

        def joined(name):
            return np.concatenate([c[name] for c in columns]) if name in columns[0] else None

        self._setup(joined("ra"), joined("dec"), patch_ids=np.repeat(np.arange(len(ids)), [len(c["ra"]) for c in columns]),
                    num_patches=len(ids), weights=joined("weights"), redshifts=joined("redshifts"),
                    cache_directory=directory, stored_meta=metas)

    @classmethod
    def _from_columns(cls, ra, dec, **kwargs):
        new = cls.__new__(cls)
        new._setup(ra, dec, **kwargs)
        return new

    def _setup(self, ra, dec, *, patch_ids, num_patches: int | None = None, weights=None, redshifts=None,
               patch_centers: AngularCoordinates | None = None, cache_directory=None, stored_meta=None,
               xyz=None) -> None:
        """Common initialiser; coordinates in radian. ``xyz`` = ``radec_to_xyz(ra, dec)`` if already known."""
        ra = np.asarray_chkfinite(ra, dtype=np.float64)
        dec = np.asarray_chkfinite(dec, dtype=np.float64)
        patch_ids = np.asarray(patch_ids)
        if not (len(ra) == len(dec) == len(patch_ids)):
            raise ValueError("input columns differ in length")
        if len(ra) == 0:
            raise ValueError("catalogue is empty")
        if patch_ids.min() < 0 or patch_ids.max() > PATCH_ID_MAX:
            raise ValueError(f"'patch_ids' must be in range [0, {PATCH_ID_MAX}]")
        patch_ids = patch_ids.astype(np.int64)
        num = int(patch_ids.max()) + 1 if num_patches is None else int(num_patches)
        _check_patch_count(num)
        if patch_ids.max() >= num:
            raise ValueError("patch id exceeds the number of patches")
        sizes = np.bincount(patch_ids, minlength=num)
        if np.any(sizes == 0):  # same restriction as the reference (catalog.py:944-947)
            empty = np.flatnonzero(sizes == 0).tolist()
            raise ValueError(f"empty patches are not supported (patch ids {empty})")
        weights = None if weights is None else np.asarray_chkfinite(weights, dtype=np.float64)
        redshifts = None if redshifts is None else np.asarray_chkfinite(redshifts, dtype=np.float64)
        # unit vectors: computed once per catalogue (assignment, patch metadata and the device layouts all use
        # these values -- the exact host numbers the pair predicate runs on)
        xyz = radec_to_xyz(ra, dec) if xyz is None else tuple(np.asarray(c, dtype=np.float64) for c in xyz)
        columns = [ra, dec, *xyz] + [c for c in (weights, redshifts) if c is not None]
        grouped = False
        if np.all(patch_ids[1:] >= patch_ids[:-1]):
            # already grouped by patch (a restored cache): the stable order is the identity. The catalogue keeps its own copy
            # of the columns all the same -- a later change of the caller's arrays must not reach into it
            columns = [np.array(c, dtype=np.float64, copy=True) for c in columns]
            grouped = True
        elif len(ra) >= HOST_GROUP_MIN:  # one threaded counting sort over all columns (yawhip_host_group_columns)
            from . import _lib

            try:
                columns, _ = _lib.group_columns(patch_ids, num, columns)
                grouped = True
            except _lib.YawhipError:  # no library on this machine: catalogue preparation is host work, numpy does it too
                pass
        if not grouped:
            order = _stable_argsort_small(patch_ids, num)
            columns = [c[order] for c in columns]
        self._ra, self._dec, self._xyz = columns[0], columns[1], tuple(columns[2:5])
        rest = iter(columns[5:])
        self._w = None if weights is None else next(rest)
        self._z = None if redshifts is None else next(rest)
        self._patch_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self._layouts: dict = {}
        self._active_layout = None
        self.cache_directory = None if cache_directory is None else Path(cache_directory)
        if patch_centers is not None and len(patch_centers) != num:
            raise ValueError("number of patch centers does not match the number of patches")

        def patch_meta(pid):
            lo, hi = int(self._patch_off[pid]), int(self._patch_off[pid + 1])
            if stored_meta and pid in stored_meta:  # meta.yml of the cache (patch.py:359-362)
                m = stored_meta[pid]
                return Metadata(num_records=int(m["num_records"]), sum_weights=float(m["sum_weights"]),
                                center=AngularCoordinates(m["center"]), radius=AngularDistances(m["radius"]))
            return Metadata.compute(
                range(hi - lo),  # only its length is used when the unit vectors are given
                weights=None if self._w is None else self._w[lo:hi],
                center=None if patch_centers is None else patch_centers[pid],
                xyz=tuple(c[lo:hi] for c in self._xyz),
            )

        if len(ra) >= _threads.MIN_PARALLEL and _threads.pool_size() > 1:  # patches are independent
            metas = list(_threads._executor().map(patch_meta, range(num)))
        else:
            metas = [patch_meta(pid) for pid in range(num)]
        self._patches = {pid: Patch(self, int(self._patch_off[pid]), int(self._patch_off[pid + 1]), metas[pid])
                         for pid in range(num)}

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_arrays(cls, ra, dec, *, weights=None, redshifts=None, patch_centers=None, patch_ids=None,
                    patch_num: int | None = None, degrees: bool = True, cache_directory=None, overwrite: bool = False,
                    probe_size: int = -1):
        """Build from plain arrays. One of ``patch_centers`` (nearest-centre assignment),
        ``patch_ids`` (pre-assigned, contiguous from 0) or ``patch_num`` (k-means) is required, with
        that precedence (``PatchMode.determine``, catalog.py:95-167)."""
        ra = np.asarray_chkfinite(ra, dtype=np.float64)
        dec = np.asarray_chkfinite(dec, dtype=np.float64)
        if degrees:  # datachunk.py:265-267
            ra, dec = np.deg2rad(ra), np.deg2rad(dec)
        centers = None
        if patch_centers is not None:
            if isinstance(patch_centers, Catalog):
                centers = patch_centers.get_centers()
            elif isinstance(patch_centers, AngularCoordinates):
                centers = patch_centers
            else:
                raise TypeError("'patch_centers' must be a set of coordinates or another catalog")
            _check_patch_count(len(centers))
        elif patch_ids is None:
            if patch_num is None:
                raise ValueError("no patch method specified")
            if not isinstance(patch_num, (int, np.integer)):
                raise TypeError("'patch_num' must be an integer")
            _check_patch_count(int(patch_num))
            if probe_size < 10 * patch_num:
                probe_size = int(100_000 * np.sqrt(patch_num))
            step = max(1, len(ra) // probe_size)
            x, y, z = radec_to_xyz(ra[::step], dec[::step])
            centers = kmeans_centers(np.column_stack([x, y, z]), None if weights is None else np.asarray(weights)[::step],
                                     int(patch_num))
        num = None
        xyz = None
        if centers is not None:
            xyz = radec_to_xyz(ra, dec)
            patch_ids = nearest_center(xyz, centers.to_3d())
            num = len(centers)
        new = cls._from_columns(ra, dec, patch_ids=patch_ids, num_patches=num, weights=weights, redshifts=redshifts,
                                patch_centers=centers, cache_directory=None, xyz=xyz)
        if cache_directory is not None:
            new.to_cache(cache_directory, overwrite=overwrite)
        return new

    @classmethod
    def from_dataframe(cls, cache_directory, dataframe, *, ra_name: str, dec_name: str, weight_name: str | None = None,
                       redshift_name: str | None = None, patch_centers=None, patch_name: str | None = None,
                       patch_num: int | None = None, kappa_name: str | None = None, degrees: bool = True,
                       overwrite: bool = False, progress: bool = False, max_workers: int | None = None,
                       chunksize: int | None = None, probe_size: int = -1, **reader_kwargs):
        """Same signature as ``yaw.Catalog.from_dataframe`` (catalog.py:980-1108). ``dataframe`` may
        be a pandas DataFrame or any mapping from column name to array. ``cache_directory`` may be
        ``None`` (nothing is written: the catalogue lives in memory and in HBM); a path gets a cache in
        the reference's on-disk format, readable by both packages (``overwrite`` as in the reference).
        ``kappa_name`` (scalar-field correlations) is outside the nn pair-count path."""
        if kappa_name is not None:
            raise NotImplementedError("scalar field ('kappa') correlations are not part of the nn pair-count path")
        if patch_name is not None and not isinstance(patch_name, str):
            raise TypeError("'patch_name' must be a string")

        def column(name):
            return None if name is None else np.asarray(dataframe[name])

        use_ids = patch_centers is None and patch_name is not None
        return cls.from_arrays(
            column(ra_name), column(dec_name), weights=column(weight_name), redshifts=column(redshift_name),
            patch_centers=patch_centers, patch_ids=column(patch_name) if use_ids else None, patch_num=patch_num,
            degrees=degrees, cache_directory=cache_directory, overwrite=overwrite, probe_size=probe_size,
        )

    @classmethod
    def from_file(cls, cache_directory, path, *, ra_name: str, dec_name: str, weight_name: str | None = None,
                  redshift_name: str | None = None, patch_centers=None, patch_name: str | None = None,
                  patch_num: int | None = None, kappa_name: str | None = None, degrees: bool = True, **kwargs):
        """Parquet (``.pqt/.parquet``) or ``.npz`` input (reference: catalog.py:1111-1243; FITS and
        HDF5 readers need libraries that are outside this build's scope)."""
        path = Path(path)
        names = [n for n in (ra_name, dec_name, weight_name, redshift_name, patch_name) if n is not None]
        if path.suffix.lower() in (".pqt", ".parquet"):
            import pyarrow.parquet as pq

            table = pq.read_table(str(path), columns=names)
            frame = {n: table[n].to_numpy() for n in names}
        elif path.suffix.lower() == ".npz":
            with np.load(str(path)) as data:
                frame = {n: data[n] for n in names}
        else:
            raise ValueError(f"unsupported file type '{path.suffix}' (supported: .pqt, .parquet, .npz)")
        return cls.from_dataframe(cache_directory, frame, ra_name=ra_name, dec_name=dec_name, weight_name=weight_name,
                                  redshift_name=redshift_name, patch_centers=patch_centers, patch_name=patch_name,
                                  patch_num=patch_num, kappa_name=kappa_name, degrees=degrees, **kwargs)

    # ------------------------------------------------------------------ mapping interface
    def __len__(self) -> int:
        return len(self._patches)

    def __getitem__(self, patch_id: int) -> Patch:
        return self._patches[patch_id]

    def __iter__(self):
        yield from sorted(self._patches)

    def __repr__(self) -> str:
        return (f"Catalog(num_patches={self.num_patches}, num_records={len(self._ra)}, "
                f"weights={self.has_weights}, redshifts={self.has_redshifts})")

    @property
    def num_patches(self) -> int:
        return len(self)

    @property
    def has_weights(self) -> bool:
        return self._w is not None

    @property
    def has_redshifts(self) -> bool:
        return self._z is not None

    def get_num_records(self) -> tuple:
        return tuple(p.meta.num_records for p in self.values())

    def get_sum_weights(self) -> tuple:
        return tuple(p.meta.sum_weights for p in self.values())

    def get_centers(self) -> AngularCoordinates:
        return AngularCoordinates.from_coords(p.meta.center for p in self.values())

    def get_radii(self) -> AngularDistances:
        return AngularDistances.from_dists(p.meta.radius for p in self.values())

    # ------------------------------------------------------------------ cache on disk
    def to_cache(self, cache_directory, *, overwrite: bool = False) -> None:
        """Write the catalogue in the reference's cache format (``patch_ids.bin``, ``patch_<id>/data.bin``,
        ``patch_<id>/meta.yml``); restore with ``Catalog(cache_directory)`` here or in the reference."""
        import shutil

        import yaml

        directory = Path(cache_directory)
        if directory.exists():
            if not overwrite:
                raise FileExistsError(f"cache directory exists and overwrite=False: {directory}")
            if any(directory.iterdir()) and not (directory / PATCH_INFO_FILE).exists():
                raise FileExistsError(f"not a catalog cache, cannot be overwritten: {directory}")
            shutil.rmtree(directory)
        directory.mkdir(parents=True)
        np.arange(self.num_patches, dtype=np.int16).tofile(directory / PATCH_INFO_FILE)
        for pid, patch in self._patches.items():
            patch_dir = directory / PATCH_NAME_TEMPLATE.format(pid)
            patch_dir.mkdir()
            lo, hi = patch._lo, patch._hi
            write_patch_file(patch_dir / PATCH_DATA_FILE, self._ra[lo:hi], self._dec[lo:hi],
                             None if self._w is None else self._w[lo:hi], None if self._z is None else self._z[lo:hi])
            with (patch_dir / PATCH_META_FILE).open("w") as f:
                yaml.safe_dump(patch.meta.to_dict(), f, indent=4)
        self.cache_directory = directory

    # ------------------------------------------------------------------ device layout
    def _unit_vectors(self):
        if self._xyz is None:
            self._xyz = radec_to_xyz(self._ra, self._dec)  # the exact host values the predicate runs on
        return self._xyz

    def build_trees(self, binning=None, *, closed=Closed.right, leafsize: int = 16, force: bool = False,
                    progress: bool = False, max_workers: int | None = None) -> PatchLayout:
        """Prepare (and cache) the (patch, bin)-sorted layout for ``binning`` (array of edges or
        ``None``).  Keeps the name and arguments of ``yaw.Catalog.build_trees`` (catalog.py:1406-1461);
        no tree is built -- ``leafsize`` is accepted and ignored."""
        bins = None if binning is None else (binning if isinstance(binning, Binning) else Binning(binning, closed=closed))
        if bins is not None and not self.has_redshifts:
            raise ValueError("patch has no 'redshifts' attached")  # trees.py:396-397
        key = None if bins is None else (bins.edges.tobytes(), str(bins.closed))
        if not force and key in self._layouts:
            self._active_layout = self._layouts[key]
            return self._active_layout
        x, y, z = self._unit_vectors()
        num_patches = self.num_patches
        if bins is None:
            layout = PatchLayout(x, y, z, self._w, self._patch_off.copy(), num_patches, 1)
        else:
            num_bins = len(bins)
            bin_idx = bins.assign(self._z)
            patch_of = np.repeat(np.arange(num_patches), np.diff(self._patch_off))
            offsets = np.zeros(num_patches * num_bins + 1, dtype=np.int64)
            columns = [x, y, z] + ([] if self._w is None else [self._w])
            grouped = False
            if len(x) >= HOST_GROUP_MIN:  # (patch, bin) grouping in one threaded pass; objects outside the binning dropped
                from . import _lib

                seg_key = np.where(bin_idx >= 0, patch_of * num_bins + bin_idx, -1)
                try:
                    columns, sizes = _lib.group_columns(seg_key, num_patches * num_bins, columns)
                    np.cumsum(sizes, out=offsets[1:])
                    grouped = True
                except _lib.YawhipError:  # no library on this machine: numpy below
                    pass
            if not grouped:
                keep = np.flatnonzero(bin_idx >= 0)
                seg_key = patch_of[keep] * num_bins + bin_idx[keep]
                order = keep[_stable_argsort_small(seg_key, num_patches * num_bins)]
                np.cumsum(np.bincount(seg_key, minlength=num_patches * num_bins), out=offsets[1:])
                columns = [c[order] for c in columns]
            layout = PatchLayout(columns[0], columns[1], columns[2], None if self._w is None else columns[3], offsets,
                                 num_patches, num_bins)
        self._layouts[key] = layout
        self._active_layout = layout  # what the next count_pairs() uses, like the cached trees.pkl
        return layout

    def drop_layouts(self) -> None:
        """Release cached layouts (and with them the device copies)."""
        for layout in self._layouts.values():
            for dev in layout.device.values():
                dev.free()
            layout.device.clear()
        self._layouts.clear()
        self._active_layout = None
