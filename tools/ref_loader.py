"""Load the read-only reference package (``/root/reference/src/yaw``) in THIS container only.

Test/fixture infrastructure, never shipped to the GPU box and never imported by the product
package.  Several optional third-party modules the reference imports at module level are not
installed here (strenum, h5py, treecorr, astropy) and ``yaw/_version.py`` is a setuptools_scm
artefact that is not in the tree; none of them is touched by the nn pair-count arithmetic, so
empty placeholder modules are registered purely so that ``import yaw`` succeeds
(recipe: SURVEY.md Appendix A).  Only angular units (rad/deg/arcmin/arcsec) are usable with
this loader because no cosmology implementation is present.
"""
from __future__ import annotations

import enum
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


# module-level (picklable: the reference's process pool pickles the Configuration with its cosmology)
class FLRW:
    pass


class _Planck15(FLRW):
    name = "Planck15"


class Quantity:
    pass


def load_reference(num_threads: int = 1):
    os.environ.setdefault("YAW_NUM_THREADS", str(num_threads))
    if "yaw" in sys.modules:
        return sys.modules["yaw"]
    if not os.path.isdir(REFERENCE_SRC):
        raise RuntimeError("reference sources not present (GPU box?) - golden fixtures only")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class StrEnum(str, enum.Enum):
        def __str__(self):
            return str(self.value)

        def _generate_next_value_(name, *a):
            return name

    mod("strenum", StrEnum=StrEnum)
    mod("h5py", File=None, Group=object)
    mod("treecorr")

    units = mod("astropy.units", Quantity=Quantity, Mpc=1.0)
    cosmo = mod(
        "astropy.cosmology",
        FLRW=FLRW,
        Planck15=_Planck15(),
        cosmology_equal=lambda a, b: a is b,
        z_at_value=None,
        available=("Planck15",),
    )
    io = mod("astropy.io", fits=None)
    mod("astropy.io.fits")
    mod("astropy", units=units, cosmology=cosmo, io=io)
    mod("yaw._version", __version__="0.0.0", __version_tuple__=(0, 0, 0))
    sys.path.insert(0, REFERENCE_SRC)
    import yaw

    return yaw
