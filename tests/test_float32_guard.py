"""The float32 guard of the band kernel (k_count_band32, yet_another_wizz_amd/csrc/yawhip.hip): for unit vectors rounded
to float32, the float32 distance s32 = fma(dz, dz, fma(dy, dy, dx * dx)) stays within
g(s) = 2.1e-7 sqrt(s) + 5e-7 s + 1e-12 of the float64 value of the parity contract. The derivation is in the kernel's
header; here the bound is checked on a few million pairs, with adversarial coordinates included (CPU, numpy)."""
import os
import re

import numpy as np

from conftest import ROOT


def _s32(a, b):
    """The kernel's float32 arithmetic, emulated: float32 images, float32 differences, mul + 2 fma."""
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    d = a32 - b32  # float32 subtraction, correctly rounded
    dx, dy, dz = (d[:, i].astype(np.float64) for i in range(3))
    acc = (dx * dx).astype(np.float32).astype(np.float64)            # v_mul_f32 (the product of two float32 is exact in float64)
    acc = (dy * dy + acc).astype(np.float32).astype(np.float64)     # v_fma_f32: one rounding of the exact sum
    return (dz * dz + acc).astype(np.float32).astype(np.float64)


def _s64(a, b):
    d = a - b
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]


def _guard(s):
    return 2.1e-7 * np.sqrt(s) + 5e-7 * s + 1e-12


def test_constants_match_the_kernel_source():
    src = open(os.path.join(ROOT, "yet_another_wizz_amd", "csrc", "yawhip.hip")).read()
    assert re.search(r"BAND32_GUARD_SQRT = 2\.1e-7;", src)
    assert src.count("BAND32_GUARD_SQRT * std::sqrt(te) + 5e-7 * te + 1e-12") == 2


def test_float32_guard_holds():
    rng = np.random.default_rng(20261004)
    worst = 0.0
    for trial in range(8):
        n = 500_000
        a = rng.normal(size=(n, 3))
        if trial % 2:  # points near the coordinate axes and planes: coordinates next to 1, 0.5, 0.25 (binade borders) and 0
            a *= np.array([1.0, 1e-3, 1e-5])[rng.permuted(np.tile(np.arange(3), (n, 1)), axis=1)]
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        v = rng.normal(size=(n, 3))
        v -= (v * a).sum(1, keepdims=True) * a
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        ang = 10.0 ** rng.uniform(-6.0, np.log10(np.pi), n)
        b = a * np.cos(ang)[:, None] + v * np.sin(ang)[:, None]
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        s64 = _s64(a, b)
        err = np.abs(_s32(a, b) - s64)
        ratio = err / _guard(s64)
        worst = max(worst, float(ratio.max()))
        assert np.all(ratio <= 1.0), (trial, float(ratio.max()))
    assert 0.05 < worst <= 1.0  # the bound is conservative, not absurdly so
