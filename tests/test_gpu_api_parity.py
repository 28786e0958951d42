"""GPU: the full drop-in API (Catalog -> crosscorrelate / autocorrelate -> CorrFunc.sample) on the
HIP path against outputs captured from the reference, plus size-independent properties at scale."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["u", "w"])
@pytest.mark.parametrize("cfg,closed", [("s2", "right"), ("s2", "left"), ("rw", "right")])
def test_full_driver_vs_reference_on_gpu(tag, cfg, closed):
    helpers.run_full_case(tag, cfg, closed)


def test_twodflens_vs_reference_on_gpu():
    helpers.run_twodflens_case()


def test_reference_end_to_end_known_answer_on_gpu():
    helpers.run_reference_example_case()


def test_native_library_is_the_one_running():
    from yet_another_wizz_amd import _lib

    assert _lib.device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libyawhip.so" in f.read()
