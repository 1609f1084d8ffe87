#!/usr/bin/env python3
"""Pins dsp_slam_amd/map_objects.py's reader against the REFERENCE'S OWN parse loop (build container only).

The reference reads MapObjects.txt inside the `__main__` block of extract_map_objects.py (lines 46-63): ids, 3x4 poses completed to 4x4
and saved as <id>.npy, codes handed to MeshExtractor.extract_mesh_from_code.  This script writes a MapObjects.txt (our writer, in the
format of src/System_util.cc:123-146 incl. Eigen's column-aligned code line), runs the UNMODIFIED reference script on it with runpy --
get_decoder / MeshExtractor / write_mesh_to_ply replaced by recorders, so no GPU, weights or scikit-image are needed -- and stores the
file's bytes next to what the reference read from it in tests/golden/golden_map_objects.npz.
"""
import os
import runpy
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from dsp_slam_amd.map_objects import write_map_objects  # noqa: E402


def main():
    ref_shim.install()
    import reconstruct.optimizer as ropt
    import reconstruct.utils as rutils
    rng = np.random.default_rng(2026)
    objs = []
    for i in (12, 3, 7, 40):
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :4] = rng.normal(size=(3, 4)).astype(np.float32) * np.float32(3.0)
        code = (rng.normal(size=64) * 0.2).astype(np.float32)
        code[5] = 0.0
        code[6] = -12.5          # a wide coefficient: Eigen pads every other one to its width
        objs.append(dict(id=i, pose=pose, code=code))
    tmp = tempfile.mkdtemp(prefix="dsp_map_")
    path = os.path.join(tmp, "MapObjects.txt")
    write_map_objects(path, objs)
    seen_codes = []

    class RecordingMeshExtractor(object):
        def __init__(self, *a, **k):
            pass

        def extract_mesh_from_code(self, code):
            seen_codes.append(np.asarray(code).copy())
            return types.SimpleNamespace(vertices=np.zeros((0, 3), np.float32), faces=np.zeros((0, 3), np.int32))

    ropt.MeshExtractor = RecordingMeshExtractor
    rutils.get_decoder = lambda cfg: None
    rutils.get_configs = lambda p: types.SimpleNamespace(optimizer=types.SimpleNamespace(code_len=64))
    rutils.write_mesh_to_ply = lambda *a, **k: None
    argv = sys.argv
    sys.argv = ["extract_map_objects.py", "-c", "unused.json", "-m", tmp, "-n", "16"]
    try:
        runpy.run_path(os.path.join(ref_shim.REFERENCE_ROOT, "extract_map_objects.py"), run_name="__main__")
    finally:
        sys.argv = argv
    ids = sorted(o["id"] for o in objs)
    poses = np.stack([np.load(os.path.join(tmp, "objects", "%d.npy" % i)) for i in ids])
    assert len(seen_codes) == len(ids)
    out = os.path.join(ROOT, "tests", "golden", "golden_map_objects.npz")
    np.savez_compressed(out, text=np.frombuffer(open(path, "rb").read(), np.uint8), ids=np.array(ids, np.int64), poses=poses,
                        codes=np.stack(seen_codes), codes_dtype=np.array(str(seen_codes[0].dtype)), poses_dtype=np.array(str(poses.dtype)))
    print("wrote", out, "ids", ids, "pose dtype", poses.dtype, "code dtype", seen_codes[0].dtype)


if __name__ == "__main__":
    main()
