"""The C-ABI library loads on a CPU-only box and exports every symbol include/yawhip.h declares.
No compute call is made here (that needs a GPU and lives in the -m gpu tests)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "yawhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yawhip_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from yet_another_wizz_amd import _lib

    assert header_symbols() == sorted(_lib.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from yet_another_wizz_amd import _lib, build

    if not os.path.exists(_lib.LIB_PATH):
        build.build_library()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), f"libyawhip.so does not export {name}"
    lib.yawhip_abi_version.restype = ctypes.c_int
    assert lib.yawhip_abi_version() == 5  # yawhip_count_pairs_dense_batch
    assert hasattr(lib, "yawhip_count_pairs_dense_batch")


def test_errors_are_reported_not_thrown():
    from yet_another_wizz_amd import _lib

    lib = _lib.load_library()
    assert lib.yawhip_device_count(None) == -1  # YAWHIP_ERR_INVALID
    assert b"NULL" in lib.yawhip_last_error()
    if _lib.device_count() == 0:
        with pytest.raises(_lib.YawhipError, match="no HIP device|NO_DEVICE|device"):
            _lib.Context(0)


def test_stats_struct_matches_header():
    """Field order/types of the ctypes mirror follow the header's struct."""
    from yet_another_wizz_amd import _lib

    text = open(os.path.join(ROOT, "include", "yawhip.h")).read()
    body = re.search(r"typedef struct yawhip_stats \{(.*?)\} yawhip_stats;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int64_t|int32_t|double)\s+(\w+);", body)
    ctype = {"int64_t": ctypes.c_int64, "int32_t": ctypes.c_int32, "double": ctypes.c_double}
    assert [(n, ctype[t]) for t, n in fields] == list(_lib._Stats._fields_)


def test_dense_request_struct_matches_header():
    """The ctypes mirror of yawhip_dense_request (the record of yawhip_count_pairs_dense_batch) follows the header: same
    fields in the same order, pointers as addresses, and the C layout's size."""
    from yet_another_wizz_amd import _lib

    text = open(os.path.join(ROOT, "include", "yawhip.h")).read()
    body = re.search(r"typedef struct yawhip_dense_request \{(.*?)\} yawhip_dense_request;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names += [n.strip(" *") for n in re.sub(r"^(const\s+)?\w+\s", "", decl).split(",")]
    assert names == [n for n, _ in _lib._DenseRequest._fields_]
    assert ctypes.sizeof(_lib._DenseRequest) == 48  # two pointers, two int32, three pointers
