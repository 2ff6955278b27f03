"""stereo_amd -- MI355X-native 3D-label stereo energy minimisation.

Host-side mirror (Python) of the reference's MATLAB solver surface
(`trws.m`, `rd.m`) over the C ABI of ``libstereo_hip.so``
(``include/stereo_hip.h``).  Names, argument meaning and error behaviour follow
the reference wrappers; arrays use MATLAB shapes (K x N, 2 x E, ...).
"""
from . import _lib
from ._lib import StereoHipError, device_count, LIB_PATH  # noqa: F401
from .trws import trws, TrwsPlan  # noqa: F401
from .rd import rd  # noqa: F401
from .dispmap import dispmap_super, dispmap_ncc, dispmap_globalstereo  # noqa: F401
from .fusion import PlaneProposal  # noqa: F401
from .segment import vgg_segment_ms, vgg_segment_gb  # noqa: F401

__all__ = ["trws", "rd", "dispmap_super", "dispmap_ncc", "dispmap_globalstereo", "TrwsPlan", "PlaneProposal", "vgg_segment_ms", "vgg_segment_gb", "StereoHipError", "device_count", "LIB_PATH"]

# Load the library (and let the HIP runtime finish its one-time initialisation, which draws from
# libc rand()) when the package is imported, not inside the first solver call: a caller that seeds
# rand() to reproduce the reference's QPBO Improve does so after the import.  Without a built
# library the import still succeeds; the first call then reports it.
try:
    _lib.lib()
except StereoHipError:
    pass
