"""Mirror of the reference's `reconstruct` package for the DeepSDF Gauss-Newton path.

Put the directory that CONTAINS this package (dsp_slam_amd/) on sys.path ahead of the reference tree and `reconstruct.optimizer`,
`reconstruct.loss`, `reconstruct.loss_utils`, `reconstruct.utils` resolve HERE -- DSP-SLAM's C++ (src/LocalMapping.cc:38-40,
src/System.cc:90-99) then drives the MI355X path unchanged -- while everything this package does not replace keeps coming from the
REFERENCE'S OWN FILES: the sequence loaders and detectors (`reconstruct/kitti_sequence.py`, `mono_sequence.py`, `detector2d.py`,
`detector3d.py`; out of scope of the hot path) are found by appending the reference's `reconstruct/` directory to
this package's `__path__`.  `src/System.cc:97` calls `reconstruct.get_sequence(...)` right after `get_decoder`, so without this the
unchanged C++ would not get past its constructor.

Where the reference checkout is looked for, in this order:
  1. `$DSP_REFERENCE_ROOT` (the directory that holds the reference's `reconstruct/`),
  2. every later `sys.path` entry (incl. "./", which System.cc:93 appends, and the working directory) that holds a
     `reconstruct/kitti_sequence.py` which is not this package.
The search runs at import and again inside get_sequence / get_detectors (C++ may extend sys.path in between).  Nothing is copied and no
reference module is shadowed by name: a module that exists in both places (optimizer, loss, loss_utils, utils) is always this package's.
"""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
_root = _os.path.dirname(_os.path.dirname(_here))
if _root not in _sys.path:      # make `import dsp_slam_amd` work when only dsp_slam_amd/ itself is on sys.path
    _sys.path.append(_root)

_REFERENCE_ONLY = ("kitti_sequence", "mono_sequence", "detector2d", "detector3d")


def reference_package_dir():
    """The reference's `reconstruct/` directory, or None when no reference checkout can be found (see the module docstring)."""
    cands = []
    env = _os.environ.get("DSP_REFERENCE_ROOT")
    if env:
        cands.append(env)
    cands += [p if p else "." for p in _sys.path]
    cands.append(_os.getcwd())
    for c in cands:
        d = _os.path.join(_os.path.abspath(c), "reconstruct")
        try:
            if _os.path.isfile(_os.path.join(d, "kitti_sequence.py")) and not _os.path.samefile(d, _here):
                return d
        except OSError:
            continue
    return None


def _extend_path():
    d = reference_package_dir()
    if d is not None and d not in __path__:
        __path__.append(d)          # AFTER this package's own directory: the mirror's modules win, the reference supplies the rest
    return d


_extend_path()


def _need_reference(what):
    if _extend_path() is None:
        raise ImportError(
            "%s is served by the reference's own reconstruct/ package (sequence loaders and detectors are not part of the MI355X hot path), "
            "but no reference checkout was found: run from the DSP-SLAM source directory, or set DSP_REFERENCE_ROOT to it" % what)


def get_detectors(configs):
    """Same dispatch as the reference's reconstruct/__init__.py:1-12; the detector classes are the reference's own modules."""
    online = configs.detect_online
    kitti = configs.data_type == "KITTI"
    if not online:
        return (None, None) if kitti else None
    _need_reference("reconstruct.get_detectors")
    from .detector2d import get_detector2d
    if kitti:
        from .detector3d import get_detector3d
        return get_detector2d(configs), get_detector3d(configs)
    return get_detector2d(configs)


def get_sequence(data_dir, configs):
    """Same dispatch as the reference's reconstruct/__init__.py:15-22 (called from src/System.cc:97); the sequence classes are the
    reference's own modules, whose frames feed the optimiser of THIS package."""
    if configs.data_type == "KITTI":
        _need_reference("reconstruct.get_sequence")
        from .kitti_sequence import KITIISequence
        return KITIISequence(data_dir, configs)
    if configs.data_type == "Redwood" or configs.data_type == "Freiburg":   # one class serves both (reference comment, :19)
        _need_reference("reconstruct.get_sequence")
        from .mono_sequence import MonoSequence
        return MonoSequence(data_dir, configs)
