"""MapObjects.txt -- the map dump DSP-SLAM writes at exit and `extract_map_objects.py` reads back.

Format (reference src/System_util.cc:123-146, extract_map_objects.py:46-63), three lines per object:
    <object id>
    <3x4 Sim(3) object->world pose, 12 numbers, row-major, 9 significant digits>
    <shape code, 64 numbers>
"""
import numpy as np


def read_map_objects(path):
    """-> list of dict(id=int, pose=(4,4) float64 Sim(3) T_world_obj, code=(C,) float32)."""
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip() != ""]
    if len(lines) % 3:
        raise ValueError("%s: expected 3 lines per object, got %d lines" % (path, len(lines)))
    out = []
    for i in range(len(lines) // 3):
        obj_id = int(lines[3 * i].strip())
        pose = np.array([float(x) for x in lines[3 * i + 1].split()], np.float64)
        if pose.size != 12:
            raise ValueError("%s: object %d pose has %d numbers" % (path, obj_id, pose.size))
        pose = np.concatenate([pose.reshape(3, 4), [[0., 0., 0., 1.]]], 0)
        code = np.array([float(x) for x in lines[3 * i + 2].split()], np.float32)
        out.append(dict(id=obj_id, pose=pose, code=code))
    return out


def write_map_objects(path, objects):
    """objects: iterable of dict(id, pose (4,4) or (3,4), code).  Laid out after System::SaveMapCurrentFrame (src/System_util.cc:123-146):
    `fixed`, setprecision(9); the pose as twelve scalars separated by single spaces; the code right-aligned to the width of the widest
    coefficient, as Eigen's default IOFormat streams a row vector (values separated by one OR MORE spaces -- the reason the reference's
    reader skips empty items, extract_map_objects.py:59-61).
    PARITY: reader-level only.  The reference's writer is C++ / Eigen and cannot be run in this image, so the exact bytes it emits are
    not pinned; what is pinned is that the reference's own parse loop (extract_map_objects.py:46-63, run through `runpy` in
    tests/test_gpu_map_tools.py) reads these files back to the values written."""
    with open(path, "w") as f:
        for o in sorted(objects, key=lambda x: x["id"]):
            f.write("%d\n" % int(o["id"]))
            p = np.asarray(o["pose"], np.float32)[:3, :4].reshape(-1)        # Eigen::Matrix4f / Vector<float,64>: float32 values
            f.write(" ".join("%.9f" % float(v) for v in p) + "\n")
            items = ["%.9f" % float(v) for v in np.asarray(o["code"], np.float32).reshape(-1)]
            width = max(len(x) for x in items) if items else 0
            f.write(" ".join(x.rjust(width) for x in items) + "\n")
