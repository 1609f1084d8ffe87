"""Mirror of the reference's `deep_sdf` package surface that the hot path touches (decoder + loader)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:      # make `import dsp_slam_amd` work when only dsp_slam_amd/ itself is on sys.path
    _sys.path.append(_root)

from . import workspace  # noqa: F401
from .deep_sdf_decoder import Decoder  # noqa: F401
