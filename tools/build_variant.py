#!/usr/bin/env python3
"""Build a VARIANT of libyawhip.so for same-box A/B runs (build container; the result travels with gpurun):

    python tools/build_variant.py diag1 -DYAW_BAND_DIAG=1      ->  yet_another_wizz_amd/build/variants/libyawhip_diag1.so
    YAW_AMD_LIB=yet_another_wizz_amd/build/variants/libyawhip_diag1.so python bench.py ...

Only the kernel translation unit is recompiled; the in-tree product library is not touched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yet_another_wizz_amd import build  # noqa: E402

tag, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(build.OBJ_DIR, "variants")
os.makedirs(out_dir, exist_ok=True)
build.build_library()  # the sort object comes from the regular build
obj = os.path.join(out_dir, f"yawhip_{tag}.o")
lib = os.path.join(out_dir, f"libyawhip_{tag}.so")
hipcc = build.hipcc_path()
subprocess.check_call([hipcc, *build.HIPCC_FLAGS, *flags, f"-I{build.INCLUDE}", f"-I{build.CSRC}", "-c", build.SOURCES[0], "-o", obj])
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj,
                       os.path.join(build.OBJ_DIR, os.path.basename(build.SOURCES[1]) + ".o")])
print(lib)
