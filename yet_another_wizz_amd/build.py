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
HEADERS = [os.path.join(ROOT, "include", "yawhip.h"), os.path.join(CSRC, "yawhip_sort.h")]
SRC = SOURCES[0]
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG_DIR, "libyawhip.so")
OBJ_DIR = os.path.join(PKG_DIR, "build")

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


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libyawhip.so cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    newest = max(os.path.getmtime(p) for p in (*SOURCES, *HEADERS))
    return os.path.getmtime(LIB) < newest


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    """Compile every source to an object (objects of unchanged sources are reused unless ``force`` or
    ``extra_flags`` are given: the rocPRIM sorts take 30 s, the kernels are what one iterates on) and link."""
    if not force and not extra_flags and not is_stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = hipcc_path()
    newest_header = max(os.path.getmtime(p) for p in HEADERS)
    objects = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objects.append(obj)
        is_kernels = src == SOURCES[0]
        flags = list(extra_flags) if is_kernels else []  # experiment flags only concern the kernels
        fresh = os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header)
        if fresh and not flags and not (force and is_kernels) and not (force == "all"):
            continue
        cmd = [hipcc, *HIPCC_FLAGS, *flags, f"-I{INCLUDE}", f"-I{CSRC}", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objects]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="all" if "--all" in sys.argv else "--force" in sys.argv, verbose=True))
