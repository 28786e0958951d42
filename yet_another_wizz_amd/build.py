"""Build libyawhip.so (HIP, gfx950 only) in-tree with hipcc.

The shared object is git-ignored but travels to the GPU box with the working-tree snapshot.
``python -m yet_another_wizz_amd.build`` rebuilds it; ``build_library()`` is what
``__graft_entry__.build()`` calls.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
SOURCES = [os.path.join(CSRC, "yawhip.hip"), os.path.join(CSRC, "yawhip_sort.hip")]
HEADERS = [os.path.join(ROOT, "include", "yawhip.h"), os.path.join(CSRC, "yawhip_sort.h"), os.path.join(CSRC, "yawhip_band32.inc")]
SRC = SOURCES[0]
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG_DIR, "libyawhip.so")
OBJ_DIR = os.path.join(PKG_DIR, "build")
FLAGS_STAMP = os.path.join(OBJ_DIR, "flags.txt")  # the flag set the present objects and library were built with

# -ffp-contract=off: the inclusion predicate must round every product and sum separately (no FMA)
HIPCC_FLAGS = [
    "-O3",
    "--offload-arch=gfx950",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def source_sha16() -> str:
    """First 16 hex digits of the SHA-256 over the kernel sources and headers: ties a measurement (profiles/pmc_traffic.json)
    to the code that produced it."""
    import hashlib

    h = hashlib.sha256()
    for path in sorted((*SOURCES, *HEADERS)):
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libyawhip.so cannot be built")


def _flag_text(extra_flags=()) -> str:
    return " ".join([*HIPCC_FLAGS, *extra_flags])


def _built_with() -> str | None:
    try:
        with open(FLAGS_STAMP) as f:
            return f.read()
    except OSError:
        return None


def is_stale(extra_flags=()) -> bool:
    """The library is missing, older than a source, or was built with another flag set (an experiment build such as
    -DYAW_BAND_DIAG must never be mistaken for the product)."""
    if not os.path.exists(LIB):
        return True
    newest = max(os.path.getmtime(p) for p in (*SOURCES, *HEADERS))
    return os.path.getmtime(LIB) < newest or _built_with() != _flag_text(extra_flags)


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    """Compile every source to an object (objects of unchanged sources are reused unless ``force`` or
    ``extra_flags`` are given: the rocPRIM sorts take 30 s, the kernels are what one iterates on) and link."""
    if not force and not is_stale(extra_flags):
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags_changed = _built_with() != _flag_text(extra_flags)
    if os.path.exists(FLAGS_STAMP):
        os.remove(FLAGS_STAMP)  # no stamp while the build is incomplete
    hipcc = hipcc_path()
    newest_header = max(os.path.getmtime(p) for p in HEADERS)
    objects, running = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objects.append(obj)
        is_kernels = src == SOURCES[0]
        flags = list(extra_flags) if is_kernels else []  # experiment flags only concern the kernels
        fresh = os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header)
        if fresh and not (is_kernels and (force or flags_changed)) and not (force == "all"):
            continue
        cmd = [hipcc, *HIPCC_FLAGS, *flags, f"-I{INCLUDE}", f"-I{CSRC}", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        running.append((cmd, subprocess.Popen(cmd)))  # the translation units compile side by side
    for cmd, proc in running:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objects]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(FLAGS_STAMP, "w") as f:
        f.write(_flag_text(extra_flags))
    return LIB


if __name__ == "__main__":
    print(build_library(force="all" if "--all" in sys.argv else "--force" in sys.argv, verbose=True))
