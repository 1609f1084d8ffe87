"""Config / decoder plumbing -- mirror of reference reconstruct/utils.py:58-163: everything the hot path and the reference's sequence loaders
import from it (the loaders themselves stay the reference's files, see reconstruct/__init__.py).  The two viewer helpers of the reference's
module (`color_table`, `set_view`: Open3D camera set-up for reconstruct_frame.py / visualize_map.py) are not restated: asking this module for
them loads the reference's own utils.py under a private name and hands its object through (module __getattr__ below)."""
import json

import numpy as np

from deep_sdf.workspace import config_decoder


def read_calib_file(filepath):
    """KITTI calibration file -> {key: float64 array} (reference utils.py:58-73, imported by kitti_sequence.py:23): `key: v0 v1 ...` per line,
    parsing stops at the first empty line, lines whose values are not all numbers (the date stamps) are skipped."""
    out = {}
    with open(filepath, "r") as f:
        for line in f:
            if line == "\n":
                break
            key, value = line.split(":", 1)
            try:
                out[key] = np.array([float(tok) for tok in value.split()])
            except ValueError:
                continue
    return out


def load_velo_scan(file):
    """Velodyne .bin -> (N, 4) float32 [x, y, z, reflectance] (reference utils.py:76-79, imported by kitti_sequence.py:23)."""
    return np.fromfile(file, dtype=np.float32).reshape((-1, 4))


_reference_utils = None


def __getattr__(name):
    """Names of the reference's reconstruct/utils.py that this module does not restate (color_table, set_view): taken from the reference's
    own file, loaded once under a private module name (it needs the reference's environment: addict, plyfile, scikit-image)."""
    global _reference_utils
    if name.startswith("__") or name not in ("color_table", "set_view"):
        raise AttributeError("module 'reconstruct.utils' has no attribute %r" % name)
    if _reference_utils is None:
        import importlib.util
        import os
        from reconstruct import reference_package_dir
        d = reference_package_dir()
        if d is None:
            raise AttributeError("reconstruct.utils.%s lives in the reference's utils.py and no reference checkout was found "
                                 "(run from the DSP-SLAM source directory or set DSP_REFERENCE_ROOT)" % name)
        spec = importlib.util.spec_from_file_location("_dsp_reference_reconstruct_utils", os.path.join(d, "utils.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _reference_utils = mod
    return getattr(_reference_utils, name)


class ForceKeyErrorDict(dict):
    """Attribute-access dict that raises KeyError on a missing key (reference utils.py:82-84, there an
    addict.Dict subclass; addict is not required here)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for a in args:
            for k, v in dict(a).items():
                self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ForceKeyErrorDict):
            v = ForceKeyErrorDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return self[k]          # KeyError on a missing key, as the reference's __missing__ does

    def __setattr__(self, k, v):
        self[k] = v

    def __missing__(self, key):
        raise KeyError(key)


def get_configs(cfg_file):
    with open(cfg_file) as f:
        cfg_dict = json.load(f)
    return ForceKeyErrorDict(**cfg_dict)


def get_decoder(configs):
    return config_decoder(configs.DeepSDF_DIR)


def create_voxel_grid(vol_dim=128, regular=False):
    """(vol_dim^3, 3) float32 sample points of MeshExtractor over [-1,1]^3, x slowest (reference utils.py:97-116).

    regular=False (default) reproduces what the reference computes under its pinned torch 1.10 (and any torch >= 1.6):
    `overall_index.long() / vol_dim` is TRUE division there, so the y index is y + z / N and the x index x + y / N + z / N^2 --
    a grid sheared by up to one voxel, in float32 arithmetic restated here operation by operation (pinned by
    tests/golden/golden_voxel_grid.npz, recorded from the reference).  regular=True is the grid the code was evidently
    meant to build (integer division).  The marching-cubes vertices are placed on the regular lattice either way, as in
    the reference (utils.py:131-138)."""
    n = np.float32(vol_dim)
    voxel_size = np.float32(2.0 / (vol_dim - 1))
    idx = np.arange(vol_dim ** 3, dtype=np.int64)
    v = np.zeros((vol_dim ** 3, 3), np.float32)
    v[:, 2] = idx % vol_dim
    if regular:
        v[:, 1] = (idx // vol_dim) % vol_dim
        v[:, 0] = (idx // vol_dim // vol_dim) % vol_dim
    else:
        q1 = idx.astype(np.float32) / n                   # LongTensor / int -> float32 true division
        v[:, 1] = np.fmod(q1, n)
        v[:, 0] = np.fmod(q1 / n, n)
    return (v * voxel_size + np.float32(-1.0)).astype(np.float32)


def convert_sdf_voxels_to_mesh(sdf_volume, engine=None):
    """Marching cubes (level 0) on a decoded (n,n,n) grid, vertices in the [-1,1]^3 frame (reference utils.py:119-140).

    The reference calls scikit-image's marching_cubes_lewiner on the CPU; here it runs on the GPU (dsp_marching_cubes) with a
    generated classic case table: same vertices (one per sign-changing grid edge, linearly interpolated), triangulation may
    differ from Lewiner's inside ambiguous cells, vertex/face order differs.  `engine`: the dsp_slam_amd Engine to run on
    (default: the engine created last, i.e. of the decoder loaded last)."""
    vol = np.ascontiguousarray(np.asarray(sdf_volume, np.float32))
    if not (vol.min() <= 0.0 <= vol.max()):
        raise ValueError("Surface level must be within volume data range.")      # scikit-image's error for an empty surface
    if engine is None:
        from dsp_slam_amd.engine import last_engine
        engine = last_engine()
    n = vol.shape[0]
    verts, faces = engine.marching_cubes(vol, level=0.0, spacing=np.float32(2.0 / (n - 1)), origin=-1.0)
    return verts, faces


def write_mesh_to_ply(v, f, ply_filename_out):
    """Binary little-endian PLY with `vertex` (x, y, z float) and `face` (list uchar int vertex_indices) elements -- the file
    the reference writes through plyfile (utils.py:143-163, plyfile's default byte order on x86/AMD64 hosts); plyfile is
    not needed here."""
    v = np.ascontiguousarray(np.asarray(v, "<f4").reshape(-1, 3))
    f = np.asarray(f, "<i4").reshape(-1, 3)
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
              "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (v.shape[0], f.shape[0]))
    faces = np.zeros(f.shape[0], dtype=[("n", "u1"), ("vertex_indices", "<i4", (3,))])
    faces["n"] = 3
    faces["vertex_indices"] = f
    with open(ply_filename_out, "wb") as out:
        out.write(header.encode("ascii"))
        out.write(v.tobytes())
        out.write(faces.tobytes())


def read_mesh_from_ply(ply_filename):
    """Inverse of write_mesh_to_ply (triangle meshes in the layout above only)."""
    with open(ply_filename, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    head = data[:end].decode("ascii").split("\n")
    if head[0] != "ply" or head[1] != "format binary_little_endian 1.0":
        raise ValueError("unsupported PLY flavour")
    nv = int([h for h in head if h.startswith("element vertex")][0].split()[-1])
    nf = int([h for h in head if h.startswith("element face")][0].split()[-1])
    v = np.frombuffer(data, "<f4", nv * 3, end).reshape(nv, 3).copy()
    faces = np.frombuffer(data, np.dtype([("n", "u1"), ("vertex_indices", "<i4", (3,))]), nf, end + nv * 12)
    if nf and not (faces["n"] == 3).all():
        raise ValueError("not a triangle mesh")
    return v, faces["vertex_indices"].astype(np.int32)
