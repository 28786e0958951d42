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
SRC = os.path.join(PKG_DIR, "csrc", "yawhip.hip")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG_DIR, "libyawhip.so")

# -ffp-contract=off: the inclusion predicate must round every product and sum separately (no FMA)
HIPCC_FLAGS = [
    "-O3",
    "--offload-arch=gfx950",
    "-std=c++17",
    "-fPIC",
    "-shared",
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
    newest = max(os.path.getmtime(p) for p in (SRC, os.path.join(INCLUDE, "yawhip.h")))
    return os.path.getmtime(LIB) < newest


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), *HIPCC_FLAGS, *extra_flags, f"-I{INCLUDE}", "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
