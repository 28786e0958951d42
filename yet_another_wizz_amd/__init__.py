"""yet_another_wizz_amd -- the angular pair-counting hot path of yet_another_wizz
(``yaw.crosscorrelate`` / ``yaw.autocorrelate`` -> ``PatchLinkage.count_pairs``) on AMD MI355X.

The public names mirror ``yaw`` (reference src/yaw/__init__.py:6-41) for the parts on that path::

    from yet_another_wizz_amd import Catalog, Configuration, crosscorrelate

    config = Configuration.create(rmin=1, rmax=10, unit="arcmin", zmin=0.1, zmax=1.0, num_bins=30)
    ref = Catalog.from_dataframe(None, df_ref, ra_name="ra", dec_name="dec", redshift_name="z", patch_num=64)
    unk = Catalog.from_dataframe(None, df_unk, ra_name="ra", dec_name="dec", patch_centers=ref)
    rnd = Catalog.from_dataframe(None, df_rnd, ra_name="ra", dec_name="dec", patch_centers=ref)
    (cf,) = crosscorrelate(config, ref, unk, unk_rand=rnd)
    w_sp = cf.sample()

The O(N^2) work runs in hand-written HIP kernels (``csrc/yawhip.hip``) behind the C ABI of
``include/yawhip.h``; there is no CPU fallback.
"""
from .binning import Binning
from .catalog import Catalog, InconsistentPatchesError, Patch
from .config import Configuration
from .coordinates import AngularCoordinates, AngularDistances
from .corrdata import CorrData, SampledData
from .corrfunc import CorrFunc
from .measurements import PatchLinkage, autocorrelate, crosscorrelate
from .paircounts import NormalisedCounts, PatchedCounts, PatchedSumWeights
from .redshifts import RedshiftData

__version__ = "0.1.0"

__all__ = [
    "AngularCoordinates",
    "AngularDistances",
    "Binning",
    "Catalog",
    "Configuration",
    "CorrData",
    "CorrFunc",
    "InconsistentPatchesError",
    "NormalisedCounts",
    "Patch",
    "PatchLinkage",
    "PatchedCounts",
    "PatchedSumWeights",
    "RedshiftData",
    "SampledData",
    "autocorrelate",
    "crosscorrelate",
]
