"""ctypes binding of ``libyawhip.so`` (C ABI: ``include/yawhip.h``).

There is deliberately no CPU fallback: if the shared object is missing or no MI355X is visible the
calls raise ``YawhipError`` -- a measurement must never silently run somewhere else.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# YAW_AMD_LIB points experiments at a variant build (tools/build_variant.py); the product loads the in-tree library
LIB_PATH = os.environ.get("YAW_AMD_LIB") or os.path.join(_PKG_DIR, "libyawhip.so")

DEFAULT_STRIP_MICRO = 5000  # the library's default strip grid spacing, in 1e-6 chord units
KERNEL_AUTO, KERNEL_EXACT, KERNEL_FILTER, KERNEL_SWEEP, KERNEL_BAND = 0, 1, 2, 3, 4
KERNEL_IDS = {"auto": KERNEL_AUTO, "exact": KERNEL_EXACT, "filter": KERNEL_FILTER, "sweep": KERNEL_SWEEP,
              "band": KERNEL_BAND}

# every symbol include/yawhip.h declares (tests check the export list against this)
ABI_SYMBOLS = (
    "yawhip_last_error",
    "yawhip_abi_version",
    "yawhip_device_count",
    "yawhip_ctx_create",
    "yawhip_ctx_create_multi",
    "yawhip_ctx_device_count",
    "yawhip_ctx_destroy",
    "yawhip_ctx_set_option",
    "yawhip_catalog_upload",
    "yawhip_catalog_upload_axis",
    "yawhip_catalog_sort_axis",
    "yawhip_catalog_free",
    "yawhip_catalog_device_bytes",
    "yawhip_count_pairs",
    "yawhip_count_pairs_dense",
    "yawhip_count_pairs_dense_batch",
    "yawhip_count_pairs_rows_device",
    "yawhip_job_work",
    "yawhip_assign_patches",
    "yawhip_host_group_columns",
    "yawhip_host_scatter_rows",
)


class YawhipError(RuntimeError):
    """Raised when libyawhip.so is unavailable or a call into it fails."""


class _Stats(ctypes.Structure):
    _fields_ = [
        ("candidate_pairs", ctypes.c_int64),
        ("evaluated_pairs", ctypes.c_int64),
        ("algorithmic_bytes", ctypes.c_int64),
        ("n_workgroups", ctypes.c_int64),
        ("n_launches", ctypes.c_int32),
        ("kernel_used", ctypes.c_int32),
        ("kernel_ms", ctypes.c_double),
        ("total_ms", ctypes.c_double),
        ("count_ms", ctypes.c_double),
        ("layout_mode", ctypes.c_int32),
        ("n_orientations", ctypes.c_int32),
        ("exact_reevaluations", ctypes.c_int64),
        ("band_variant", ctypes.c_int32),
        ("merged_triples", ctypes.c_int32),
    ]


class _DenseRequest(ctypes.Structure):
    """``yawhip_dense_request`` of include/yawhip.h (pointers as plain addresses)."""
    _fields_ = [
        ("c1", ctypes.c_void_p),
        ("c2", ctypes.c_void_p),
        ("n_jobs", ctypes.c_int32),
        ("halve_diagonal", ctypes.c_int32),
        ("jobs", ctypes.c_void_p),
        ("dense", ctypes.c_void_p),
        ("stats", ctypes.c_void_p),
    ]


@dataclass
class CountStats:
    candidate_pairs: int = 0
    evaluated_pairs: int = 0
    algorithmic_bytes: int = 0
    n_workgroups: int = 0
    n_launches: int = 0
    kernel_used: int = 0
    kernel_ms: float = 0.0
    total_ms: float = 0.0
    count_ms: float = 0.0
    layout_mode: int = 0
    n_orientations: int = 0
    exact_reevaluations: int = 0
    band_variant: int = 0
    merged_triples: int = 0


_dp = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p
_lib = None


def load_library() -> ctypes.CDLL:
    """Load libyawhip.so and declare its prototypes. Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YawhipError(
            f"{LIB_PATH} not found: build it with `python -m yet_another_wizz_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the pair-count path."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as err:  # e.g. libamdhip64 missing
        raise YawhipError(f"cannot load {LIB_PATH}: {err}") from err
    lib.yawhip_last_error.restype = ctypes.c_char_p
    lib.yawhip_last_error.argtypes = []
    lib.yawhip_abi_version.restype = ctypes.c_int
    lib.yawhip_abi_version.argtypes = []
    lib.yawhip_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    lib.yawhip_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]
    lib.yawhip_ctx_create_multi.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(_vp)]
    lib.yawhip_ctx_device_count.argtypes = [_vp, ctypes.POINTER(ctypes.c_int)]
    lib.yawhip_ctx_destroy.argtypes = [_vp]
    lib.yawhip_ctx_set_option.argtypes = [_vp, ctypes.c_char_p, ctypes.c_int64]
    lib.yawhip_catalog_upload.argtypes = [
        _vp, ctypes.c_int64, _dp, _dp, _dp, _dp, ctypes.c_int32, ctypes.c_int32, _i64p, ctypes.POINTER(_vp),
    ]
    lib.yawhip_catalog_upload_axis.argtypes = [
        _vp, ctypes.c_int64, _dp, _dp, _dp, _dp, ctypes.c_int32, ctypes.c_int32, _i64p, ctypes.c_int32, ctypes.POINTER(_vp),
    ]
    lib.yawhip_catalog_sort_axis.argtypes = [_vp, ctypes.POINTER(ctypes.c_int32)]
    lib.yawhip_catalog_free.argtypes = [_vp]
    lib.yawhip_catalog_device_bytes.argtypes = [_vp, _i64p]
    lib.yawhip_count_pairs.argtypes = [
        _vp, _vp, _vp, ctypes.c_int32, _i32p, ctypes.c_int32, ctypes.c_int32, _dp, ctypes.c_int32,
        _i64p, _dp, ctypes.POINTER(_Stats),
    ]
    lib.yawhip_count_pairs_dense.argtypes = [  # array arguments as plain addresses (_addr): this is the per-call hot path
        _vp, _vp, _vp, ctypes.c_int32, _vp, ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32,
        ctypes.c_int32, _vp, _vp, ctypes.c_int32, _vp, ctypes.POINTER(_Stats),
    ]
    lib.yawhip_count_pairs_dense_batch.argtypes = [
        _vp, ctypes.c_int32, ctypes.POINTER(_DenseRequest), ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, ctypes.c_int32,
        _vp, _vp,
    ]
    lib.yawhip_count_pairs_rows_device.argtypes = [
        _vp, _vp, _vp, ctypes.c_int32, _i32p, ctypes.c_int32, ctypes.c_int32, _dp, ctypes.c_int32,
        ctypes.c_int64, _i32p, ctypes.POINTER(_vp), ctypes.POINTER(_Stats),
    ]
    lib.yawhip_assign_patches.argtypes = [_vp, ctypes.c_int64, _dp, _dp, _dp, ctypes.c_int32, _dp, _i32p]
    lib.yawhip_job_work.argtypes = [
        _vp, _vp, _vp, ctypes.c_int32, _i32p, ctypes.c_int32, ctypes.c_int32, _dp, ctypes.c_int32, _i64p,
    ]
    lib.yawhip_host_group_columns.argtypes = [
        ctypes.c_int64, _vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(_dp), ctypes.POINTER(_dp), _i64p,
        ctypes.c_int32,
    ]
    lib.yawhip_host_scatter_rows.argtypes = [ctypes.c_int64, ctypes.c_int64, _vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64,
                                             ctypes.c_int64, _vp]
    for name in ABI_SYMBOLS:
        fn = getattr(lib, name)
        if name != "yawhip_last_error":
            fn.restype = ctypes.c_int
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().yawhip_last_error()
        raise YawhipError(f"{what} failed (status {rc}): {msg.decode() if msg else 'unknown error'}")


def device_count() -> int:
    n = ctypes.c_int(0)
    rc = load_library().yawhip_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _addr(a, dtype):
    """Address of a C-contiguous array of ``dtype`` for a ``c_void_p`` argument (half the cost of ``data_as``)."""
    if a is None:
        return None
    if a.dtype != dtype or not a.flags.c_contiguous:
        raise TypeError(f"expected a C-contiguous {np.dtype(dtype).name} array, got {a.dtype} (contiguous={a.flags.c_contiguous})")
    return a.ctypes.data


class Context:
    """One GPU (one HIP stream), or several GPUs of the node behind one handle (``device`` a sequence of ids:
    catalogues are replicated, ``count_pairs`` splits its jobs over the devices). ``yawhip_ctx`` of include/yawhip.h."""

    def __init__(self, device=0):
        self._h = _vp()
        if isinstance(device, (list, tuple)):
            ids = (ctypes.c_int * len(device))(*[int(d) for d in device])
            _check(load_library().yawhip_ctx_create_multi(ids, len(device), ctypes.byref(self._h)), "yawhip_ctx_create_multi")
            self.devices = tuple(int(d) for d in device)
        else:
            _check(load_library().yawhip_ctx_create(int(device), ctypes.byref(self._h)), "yawhip_ctx_create")
            self.devices = (int(device),)
        self.device = self.devices[0]
        self.strip_micro = DEFAULT_STRIP_MICRO

    def set_option(self, key: str, value: int) -> None:
        _check(load_library().yawhip_ctx_set_option(self._h, key.encode(), int(value)), "yawhip_ctx_set_option")
        if key == "strip_width_micro":
            self.strip_micro = int(value)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            load_library().yawhip_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceCatalog:
    """SoA catalogue resident in HBM, sorted by (patch, bin). ``yawhip_catalog``."""

    def __init__(self, ctx: Context, x, y, z, w, n_patches: int, n_bins_or_1: int, offsets, sort_axis: int = 2,
                 strip_micro: int | None = None):
        """``strip_micro``: spacing of the strip grid of the cross-correlation layout in 1e-6 chord
        units (0 = no strips, None = whatever the context is set to)."""
        x, y, z, w = _f64(x), _f64(y), _f64(z), _f64(w)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(x)
        if not (len(y) == n and len(z) == n and (w is None or len(w) == n)):
            raise ValueError("catalogue columns differ in length")
        if len(offsets) != n_patches * n_bins_or_1 + 1:
            raise ValueError("offsets must have n_patches * n_bins_or_1 + 1 entries")
        self.ctx = ctx  # keep the context alive
        self.n, self.n_patches, self.n_bins = n, int(n_patches), int(n_bins_or_1)
        self.weighted = w is not None
        self._h = _vp()
        if strip_micro is not None:
            ctx.set_option("strip_width_micro", int(strip_micro))
            ctx.strip_micro = int(strip_micro)
        self.strip_micro = ctx.strip_micro
        _check(
            load_library().yawhip_catalog_upload_axis(
                ctx._h, n, _ptr(x, _dp), _ptr(y, _dp), _ptr(z, _dp), _ptr(w, _dp), self.n_patches, self.n_bins,
                _ptr(offsets, _i64p), int(sort_axis), ctypes.byref(self._h),
            ),
            "yawhip_catalog_upload_axis",
        )
        self.sort_axis = int(sort_axis)

    @property
    def device_bytes(self) -> int:
        b = ctypes.c_int64(0)
        _check(load_library().yawhip_catalog_device_bytes(self._h, ctypes.byref(b)), "yawhip_catalog_device_bytes")
        return b.value

    def free(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            load_library().yawhip_catalog_free(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def count_pairs(ctx: Context, c1: DeviceCatalog, c2: DeviceCatalog, jobs, thresholds, *, kernel="auto",
                want_counts=None, want_sums=None):
    """Run ``yawhip_count_pairs``.

    jobs: int[n_jobs,2]; thresholds: f64[B,E]. Returns (counts int64[n_jobs,B,E-1] | None,
    sums f64[n_jobs,B,E-1] | None, CountStats)."""
    jobs = np.ascontiguousarray(jobs, dtype=np.int32).reshape(-1, 2)
    t = np.ascontiguousarray(thresholds, dtype=np.float64)
    if t.ndim != 2:
        raise ValueError("thresholds must be [n_bins, n_edges]")
    n_bins, n_edges = t.shape
    weighted = c1.weighted or c2.weighted
    if want_counts is None:
        want_counts = not weighted
    if want_sums is None:
        want_sums = weighted
    shape = (len(jobs), n_bins, max(n_edges - 1, 0))
    counts = np.empty(shape, dtype=np.int64) if want_counts else None   # the library writes every element it is asked for
    sums = np.empty(shape, dtype=np.float64) if want_sums else None
    st = _Stats()
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
    _check(
        load_library().yawhip_count_pairs(
            ctx._h, c1._h, c2._h, len(jobs), _ptr(jobs, _i32p), n_bins, n_edges, _ptr(t, _dp), kid,
            _ptr(counts, _i64p), _ptr(sums, _dp), ctypes.byref(st),
        ),
        "yawhip_count_pairs",
    )
    stats = CountStats(**{f: getattr(st, f) for f, _ in _Stats._fields_})
    return counts, sums, stats


def count_pairs_dense(ctx: Context, c1: DeviceCatalog, c2: DeviceCatalog, jobs, thresholds, slices, fine_factors, halve_diagonal,
                      *, kernel="auto"):
    """Run ``yawhip_count_pairs_dense``: the result tensor f64[S, B, P, P] of ``PatchLinkage.count_pairs`` in one call.

    jobs: int32[n_jobs, 2] (C contiguous); thresholds: f64[B, E]; slices: int32[B, S, 2]; fine_factors: f64[B, E-1] | None."""
    n_bins, n_edges = thresholds.shape
    n_scales = slices.shape[1]
    dense = np.empty((n_scales, n_bins, c1.n_patches, c1.n_patches), dtype=np.float64)  # the library writes every element
    st = _Stats()
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
    _check(
        load_library().yawhip_count_pairs_dense(
            ctx._h, c1._h, c2._h, len(jobs), _addr(jobs, np.int32), n_bins, n_edges, _addr(thresholds, np.float64), kid,
            n_scales, _addr(slices, np.int32), _addr(fine_factors, np.float64), 1 if halve_diagonal else 0, dense.ctypes.data,
            ctypes.byref(st),
        ),
        "yawhip_count_pairs_dense",
    )
    return dense, CountStats(**{f: getattr(st, f) for f, _ in _Stats._fields_})


def count_pairs_dense_batch(ctx: Context, requests, thresholds, slices, fine_factors, *, kernel="auto"):
    """Run ``yawhip_count_pairs_dense_batch``: several counts of one measurement (same thresholds and recombination) on the
    stream at once. ``requests``: sequence of ``(c1, c2, jobs int32[n, 2], halve_diagonal)``. Returns a list of
    ``(dense f64[S, B, P, P], CountStats)`` in the order of the requests."""
    n_bins, n_edges = thresholds.shape
    n_scales = slices.shape[1]
    reqs = (_DenseRequest * len(requests))()
    outs, stats, keep = [], [], []
    for i, (c1, c2, jobs, halve) in enumerate(requests):
        jobs = np.ascontiguousarray(jobs, dtype=np.int32).reshape(-1, 2)
        dense = np.empty((n_scales, n_bins, c1.n_patches, c1.n_patches), dtype=np.float64)  # the library writes every element
        st = _Stats()
        keep.append(jobs)
        outs.append(dense)
        stats.append(st)
        reqs[i].c1, reqs[i].c2 = c1._h, c2._h
        reqs[i].n_jobs, reqs[i].halve_diagonal = len(jobs), 1 if halve else 0
        reqs[i].jobs, reqs[i].dense = jobs.ctypes.data, dense.ctypes.data
        reqs[i].stats = ctypes.addressof(st)
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
    _check(
        load_library().yawhip_count_pairs_dense_batch(
            ctx._h, len(requests), reqs, n_bins, n_edges, _addr(thresholds, np.float64), kid, n_scales,
            _addr(slices, np.int32), _addr(fine_factors, np.float64)),
        "yawhip_count_pairs_dense_batch",
    )
    return [(d, CountStats(**{f: getattr(st, f) for f, _ in _Stats._fields_})) for d, st in zip(outs, stats)]


class DeviceRows:
    """float64[n] in the HBM of ``device``, owned by a context (valid until its next call); exposes
    ``__cuda_array_interface__`` so that torch wraps it without a copy (``torch.as_tensor(rows, device=...)``)."""

    def __init__(self, ptr: int, n: int, device: int):
        self.ptr, self.n, self.device = int(ptr), int(n), int(device)
        self.__cuda_array_interface__ = {"shape": (self.n,), "typestr": "<f8", "data": (self.ptr, False), "version": 3,
                                         "strides": None}


def count_pairs_rows_device(ctx: Context, c1: DeviceCatalog, c2: DeviceCatalog, jobs, thresholds, n_rows_total: int, row_index,
                            *, kernel="auto"):
    """Run ``yawhip_count_pairs_rows_device``: this rank's rows of a sharded count, in place in the full tensor, left on
    the device. Returns (DeviceRows of n_rows_total * B * (E-1) + 1 values, CountStats)."""
    jobs = np.ascontiguousarray(jobs, dtype=np.int32).reshape(-1, 2)
    t = np.ascontiguousarray(thresholds, dtype=np.float64)
    row_index = np.ascontiguousarray(row_index, dtype=np.int32)
    n_bins, n_edges = t.shape
    st = _Stats()
    out = _vp()
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
    _check(
        load_library().yawhip_count_pairs_rows_device(
            ctx._h, c1._h, c2._h, len(jobs), _ptr(jobs, _i32p), n_bins, n_edges, _ptr(t, _dp), kid,
            int(n_rows_total), _ptr(row_index, _i32p), ctypes.byref(out), ctypes.byref(st),
        ),
        "yawhip_count_pairs_rows_device",
    )
    n = int(n_rows_total) * n_bins * (n_edges - 1) + 1
    return DeviceRows(out.value, n, ctx.device), CountStats(**{f: getattr(st, f) for f, _ in _Stats._fields_})


def job_work(ctx: Context, c1: DeviceCatalog, c2: DeviceCatalog, jobs, thresholds, *, kernel="auto") -> np.ndarray:
    """Run ``yawhip_job_work``: evaluated pair distances per job (int64[n_jobs]), nothing is counted."""
    jobs = np.ascontiguousarray(jobs, dtype=np.int32).reshape(-1, 2)
    t = np.ascontiguousarray(thresholds, dtype=np.float64)
    work = np.zeros(len(jobs), dtype=np.int64)
    kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
    _check(
        load_library().yawhip_job_work(ctx._h, c1._h, c2._h, len(jobs), _ptr(jobs, _i32p), t.shape[0], t.shape[1],
                                       _ptr(t, _dp), kid, _ptr(work, _i64p)),
        "yawhip_job_work",
    )
    return work


def group_columns(keys, num_groups: int, columns, n_threads: int = 0):
    """``yawhip_host_group_columns``: stable grouping of float64 columns by integer key (host threads, no device).
    Returns ``(grouped columns, sizes)``; entries with a negative key are dropped. Equivalent to
    ``order = np.flatnonzero(keys >= 0)[np.argsort(keys[keys >= 0], kind="stable")]; [c[order] for c in columns]``."""
    keys = np.ascontiguousarray(keys)
    if keys.dtype not in (np.dtype(np.int32), np.dtype(np.int64)):
        keys = keys.astype(np.int64)
    cols = [_f64(c) for c in columns]
    n = len(keys)
    if any(len(c) != n for c in cols):
        raise ValueError("columns and keys differ in length")
    outs = [np.empty(n, dtype=np.float64) for _ in cols]
    sizes = np.zeros(int(num_groups), dtype=np.int64)
    arr = _dp * max(len(cols), 1)
    _check(
        load_library().yawhip_host_group_columns(
            n, keys.ctypes.data_as(_vp), keys.dtype.itemsize, int(num_groups), len(cols),
            arr(*[_ptr(c, _dp) for c in cols]), arr(*[_ptr(o, _dp) for o in outs]), _ptr(sizes, _i64p), int(n_threads)),
        "yawhip_host_group_columns",
    )
    kept = int(sizes.sum())
    return [o[:kept] for o in outs], sizes


def scatter_rows(shape, cols, vals, col_factor=None) -> np.ndarray:
    """``yawhip_host_scatter_rows``: zeros(shape) viewed as [rows, row_len] with ``out[r, cols[j]] = vals[..., j]``
    (times ``col_factor[j]``). ``cols`` int64[n]; ``vals`` float64[..., n], its leading axes are the rows (any strides
    that make the rows equidistant, e.g. a transposed view of the job-major device result)."""
    out = np.empty(shape, dtype=np.float64)
    n_cols = len(cols)
    if n_cols == 0 or vals.size == 0:
        out[...] = 0.0
        return out
    vals2 = vals.reshape(-1, n_cols) if vals.ndim != 2 else vals  # a view whenever the strides allow it
    if vals2.dtype != np.float64 or vals2.strides[0] % 8 or vals2.strides[1] % 8:
        vals2 = np.ascontiguousarray(vals2, dtype=np.float64)
    n_rows = vals2.shape[0]
    if out.size % n_rows or cols.dtype != np.int64 or not cols.flags.c_contiguous:
        raise ValueError("scatter_rows: shape / cols do not fit the values")
    rc = load_library().yawhip_host_scatter_rows(
        n_rows, out.size // n_rows, out.ctypes.data, n_cols, cols.ctypes.data, vals2.ctypes.data,
        vals2.strides[0] // 8, vals2.strides[1] // 8, None if col_factor is None else col_factor.ctypes.data)
    _check(rc, "yawhip_host_scatter_rows")
    return out


def assign_patches(ctx: Context, x, y, z, centers_xyz) -> np.ndarray:
    """Run ``yawhip_assign_patches``: index of the nearest centre for every object (int32[n])."""
    x, y, z = _f64(x), _f64(y), _f64(z)
    centers = np.ascontiguousarray(centers_xyz, dtype=np.float64).reshape(-1, 3)
    out = np.empty(len(x), dtype=np.int32)
    _check(
        load_library().yawhip_assign_patches(ctx._h, len(x), _ptr(x, _dp), _ptr(y, _dp), _ptr(z, _dp), len(centers),
                                             _ptr(centers, _dp), _ptr(out, _i32p)),
        "yawhip_assign_patches",
    )
    return out
