"""Optimizer / MeshExtractor -- mirror of reference reconstruct/optimizer.py:26-223.

Same constructor arguments, method names, argument meaning and result fields as the reference, so that
DSP-SLAM's C++ (src/LocalMapping.cc:38-40; src/LocalMapping_util.cc:109-110,179-196,391-426) can call it
unchanged.  The whole Gauss-Newton loop (all iterations) runs on the MI355X inside libdspgn with no host
round trip; `reconstruct_objects` / `estimate_poses_cam_obj` are the batched forms (new, for many
independent objects per call).
"""
import numpy as np
import torch

from reconstruct.utils import ForceKeyErrorDict, create_voxel_grid, convert_sdf_voxels_to_mesh
from reconstruct.loss_utils import get_time
from dsp_slam_amd import engine as _engine
from dsp_slam_amd import _lib as _L


def _f32(a):
    """Eigen hands over Fortran-ordered float32 copies (pybind11 eigen caster); accept any strides / dtype."""
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


class Optimizer(object):
    def __init__(self, decoder, configs):
        # Attribute names are the reference's (optimizer.py:27-43): C++ reads `code_len` (LocalMapping_util.cc:413), scripts read the rest.
        self.decoder = decoder
        optim_cfg = configs.optimizer
        joint = optim_cfg.joint_optim
        for attr, key in (("k1", "k1"), ("k2", "k2"), ("k3", "k3"), ("k4", "k4"), ("b1", "b1"), ("b2", "b2"), ("lr", "learning_rate"),
                          ("s_damp", "scale_damping"), ("num_iterations_joint_optim", "num_iterations")):
            setattr(self, attr, joint[key])            # KeyError on a missing key, as attribute access on the reference's config dict (utils.py:82-84)
        for attr, key in (("code_len", "code_len"), ("num_depth_samples", "num_depth_samples"), ("cut_off", "cut_off_threshold")):
            setattr(self, attr, optim_cfg[key])
        # only the KITTI configuration carries a pose-only block; elsewhere the reference hard-codes five iterations (:41-43)
        self.num_iterations_pose_only = optim_cfg.pose_only_optim.num_iterations if configs.data_type == "KITTI" else 5
        if self.code_len not in (32, _L.CODE_LEN):      # the two code lengths the reference's C++ casts (LocalMapping_util.cc:413-423)
            raise NotImplementedError("the MI355X decoder kernels are built for 64-D and 32-D codes (got %d)" % self.code_len)
        if decoder is not None and self.code_len != getattr(decoder, "latent_size", self.code_len):
            raise ValueError("optimizer.code_len (%d) does not match the decoder's CodeLength (%d)" % (self.code_len, decoder.latent_size))
        self.verbose = True
        # ADDITION (not in the reference's configs): "compute_dtype": "f16" | "bf16" under "optimizer" opts into the low-precision compute mode
        # (dsp_batch_set_compute: 16-bit MFMA operands, fp32 accumulation -- NOT the parity path; include/dsp_gn.h).  Absent: fp32.
        try:
            dt = optim_cfg["compute_dtype"] if "compute_dtype" in optim_cfg else "f32"
        except TypeError:
            dt = "f32"
        if dt not in ("f32", "f16", "bf16"):
            raise ValueError("optimizer.compute_dtype must be f32, f16 or bf16 (got %r)" % (dt,))
        self.compute = {"f32": _L.COMPUTE_F32, "f16": _L.COMPUTE_F16, "bf16": _L.COMPUTE_BF16}[dt]

    def _params(self):
        return _engine.gn_params(self.k1, self.k2, self.k3, self.k4, self.b1, self.b2, self.lr, self.s_damp,
                                 self.num_iterations_joint_optim, self.num_depth_samples, self.cut_off,
                                 self.num_iterations_pose_only)

    # ---- pose only (reference optimizer.py:45-86) ---------------------------------------------------
    def estimate_poses_cam_obj(self, t_co_se3_list, scales, pts_list, codes):
        """Batched form: lists of per-object inputs -> (B,4,4) float32 array of optimised SE(3) poses."""
        return self.decoder.engine.estimate_pose_batch(self._params(), [_f32(t) for t in t_co_se3_list], scales,
                                                       [_f32(p) for p in pts_list], [_f32(c) for c in codes])

    def estimate_pose_cam_obj(self, t_co_se3, scale, pts, code):
        """Pose-only refinement of one detection (reference optimizer.py:45-86; called from LocalMapping_util.cc:109-110).
        t_co_se3: (4, 4) rigid object-to-camera guess; scale: the object's scale (float); pts: (M, 3) surface points in the camera frame;
        code: the object's shape code.  Returns the refined rigid object-to-camera matrix as a (4, 4) CPU torch.Tensor, like the reference."""
        out = self.estimate_poses_cam_obj([t_co_se3], [float(scale)], [pts], [code])
        return torch.from_numpy(out[0].copy())

    # ---- joint shape + pose (reference optimizer.py:88-203) -----------------------------------------
    def reconstruct_objects(self, t_cam_obj_list, pts_list, rays_list, depth_list, codes=None):
        """Batched form: B independent objects in one device run -> list of result dicts."""
        B = len(pts_list)
        codes_in = None
        if codes is not None:
            codes_in = [np.zeros(self.code_len, np.float32) if c is None else _f32(c)[:self.code_len] for c in codes]
        t, code, loss, status = self.decoder.engine.reconstruct_batch(
            self._params(), [_f32(x) for x in t_cam_obj_list], [_f32(p) for p in pts_list],
            [_f32(r) for r in rays_list], [_f32(d).reshape(-1) for d in depth_list], codes_in, compute=self.compute)
        out = []
        for i in range(B):
            if status[i] == _L.OBJ_GOOD:
                out.append(ForceKeyErrorDict(t_cam_obj=t[i].copy(), code=code[i].copy(), is_good=True,
                                             loss=torch.tensor(float(loss[i]))))
            else:   # reference: t_cam_obj=None, code=None, is_good=False, loss=<last computed loss> (:131,136,143,150)
                out.append(ForceKeyErrorDict(t_cam_obj=None, code=None, is_good=False, loss=float(loss[i])))
        return out

    def reconstruct_object(self, t_cam_obj, pts, rays, depth, code=None):
        """Joint shape + pose optimisation of one object (reference optimizer.py:88-203; LocalMapping_util.cc:179-180,391-392,402-403).
        t_cam_obj: (4, 4) Sim(3) object-to-camera start; pts: (M, 3) surface points in the camera frame; rays: (R, 3) ray directions, the
        first len(depth) of them foreground; depth: observed depth of the foreground rays (KITTI: one per surface point); code: optional
        start code (zeros when None).  Returns the reference's result dict: t_cam_obj, code, is_good, loss -- attribute access, KeyError on
        anything else."""
        start = get_time()
        rst = self.reconstruct_objects([t_cam_obj], [pts], [rays], [depth], None if code is None else [code])[0]
        if self.verbose and rst.is_good:
            print("Reconstruction takes %f seconds" % (get_time() - start))
        return rst

    @staticmethod
    def get_shape_code(result):
        """Shape code of a reconstruction result (the C++ side keeps it in MapObject::GetShapeCode,
        src/MapObject.cc:469-473).  Addition named by BASELINE.json; not present in the reference's Python."""
        return result.code


class MeshExtractor(object):
    def __init__(self, decoder, code_len=64, voxels_dim=64, regular_grid=False):
        """regular_grid=False samples the SDF exactly where the reference does (its grid is sheared by a true-division quirk,
        see reconstruct.utils.create_voxel_grid); True samples the regular lattice.  (Addition; the reference has no such switch.)"""
        self.decoder = decoder
        self.code_len = code_len
        self.voxels_dim = voxels_dim
        self.regular_grid = bool(regular_grid)
        self.voxel_points = create_voxel_grid(vol_dim=self.voxels_dim, regular=self.regular_grid)

    def decode_grid(self, code):
        """SDF on the voxels_dim^3 grid, decoded on the GPU (the part of extract_mesh_from_code that is
        decoder work, reference optimizer.py:217-218)."""
        sdf = self.decoder.engine.decode_sdf(_f32(code)[:self.code_len], self.voxel_points)
        return sdf.reshape(self.voxels_dim, self.voxels_dim, self.voxels_dim)

    def decode_grids(self, codes):
        """Batched grid decode: (n, code_len) codes -> (n, D, D, D) SDF volumes in one kernel launch (the loop of
        extract_map_objects.py:46-63 over a whole map)."""
        codes = np.stack([_f32(c)[:self.code_len] for c in codes])
        sdf = self.decoder.engine.decode_sdf_multi(codes, self.voxel_points)
        return sdf.reshape(codes.shape[0], self.voxels_dim, self.voxels_dim, self.voxels_dim)

    def extract_mesh_from_code(self, code):
        """Grid decode + marching cubes, both on the GPU without the volume leaving HBM (reference optimizer.py:214-223;
        there: GPU decode, then scikit-image marching cubes on the CPU)."""
        start = get_time()
        vertices, faces = self.decoder.engine.extract_mesh(_f32(code)[:self.code_len], self.voxels_dim, regular_grid=self.regular_grid)
        if vertices.shape[0] == 0:
            raise ValueError("Surface level must be within volume data range.")   # what scikit-image raises in the reference
        print("Extract mesh takes %f seconds" % (get_time() - start))
        return ForceKeyErrorDict(vertices=vertices, faces=faces)
