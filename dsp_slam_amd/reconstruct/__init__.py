"""Mirror of the reference's `reconstruct` package for the DeepSDF Gauss-Newton path.

Put the directory that CONTAINS this package (dsp_slam_amd/) on sys.path ahead of the reference tree and
`import reconstruct.optimizer` / `reconstruct.utils` resolve here, so DSP-SLAM's C++ (src/LocalMapping.cc:38-40,
src/System.cc:90-99) drives the MI355X path unchanged -- see INTEGRATION.md.
"""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:      # make `import dsp_slam_amd` work when only dsp_slam_amd/ itself is on sys.path
    _sys.path.append(_root)



def get_detectors(configs):
    """reference reconstruct/__init__.py:1-12 -- detectors are outside the accelerated path."""
    raise NotImplementedError("MaskRCNN / PointPillars detectors are out of scope of the MI355X hot path; "
                              "keep using the reference's reconstruct.detector2d / detector3d")


def get_sequence(data_dir, configs):
    """reference reconstruct/__init__.py:15-22 -- dataset front-ends are outside the accelerated path."""
    raise NotImplementedError("KITTI / Redwood / Freiburg sequence loaders are out of scope of the MI355X hot path; "
                              "keep using the reference's reconstruct.kitti_sequence / mono_sequence")
