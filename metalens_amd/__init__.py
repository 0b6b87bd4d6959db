"""metalens_amd - MI355X-native near-field synthesis and near-to-far-field
transform for the sbyrnes321/metalens design flow.

Host code in Python (mirroring the reference's function and class names), compute
in hand-written HIP for gfx950 behind the C ABI of include/metalens_hip.h.
"""
from . import constants, grating, interp, layout, lens_center, postprocess, synthetic  # noqa: F401
from .grating import Grating, GratingCollection  # noqa: F401
from .lens_center import HexGridSet  # noqa: F401
from .nearfield import build_nearfield, build_nearfield_big, good_fft_number  # noqa: F401
from .pipeline import HotPath  # noqa: F401
from .prepared import PreparedLens  # noqa: F401
from .sweep import SourceSweep, WavelengthSweep  # noqa: F401
from .nearfield_farfield import (FarfieldTransform, farfield_direct,  # noqa: F401
                                 farfield_from_nearfield, farfield_from_resident_nearfield,
                                 fft_direction_cosines)
