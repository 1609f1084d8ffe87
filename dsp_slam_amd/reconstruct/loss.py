"""Residual terms -- mirror of reference reconstruct/loss.py (same names, argument meaning, shapes, None-returns).

Each call runs the GPU pipeline of one Gauss-Newton linearisation (libdspgn: sampling, decoder K1/K2, per-ray scan,
J rows) for a single object and returns CPU torch tensors shaped like the reference's.
"""
import numpy as np
import torch

from reconstruct.loss_utils import _np


def compute_sdf_loss(decoder, pts_surface_cam, t_obj_cam, latent_vector):
    """reference loss.py:22-43 -> Jacobian wrt pose (N,1,7), wrt code (N,1,code_len), residuals (N,1,1)."""
    j7, jc, r = decoder.engine.compute_sdf_loss(_np(pts_surface_cam), _np(t_obj_cam), _np(latent_vector))
    n = r.shape[0]
    return (torch.from_numpy(j7).view(n, 1, 7), torch.from_numpy(jc).view(n, 1, -1), torch.from_numpy(r).view(n, 1, 1))


def compute_render_loss(decoder, ray_directions, depth_obs, t_obj_cam, sampled_ray_depth, latent_vector, th=0.01):
    """reference loss.py:46-152 -> (K,1,7), (K,1,code_len), (K,1,1), or None when < 10 samples fall in the unit sphere."""
    out, _ = decoder.engine.compute_render_loss(_np(ray_directions), _np(depth_obs), _np(t_obj_cam),
                                                _np(sampled_ray_depth), _np(latent_vector), th)
    if out is None:
        return None
    j7, jc, r = out
    k = r.shape[0]
    return (torch.from_numpy(j7).view(k, 1, 7), torch.from_numpy(jc).view(k, 1, -1), torch.from_numpy(r).view(k, 1, 1))


def compute_rotation_loss_sim3(t_obj_cam):
    """reference loss.py:155-178 (host float32; the optimiser uses the device version in gn_kernels.hip)."""
    t_oc = _np(t_obj_cam).astype(np.float32)
    t_co = np.linalg.inv(t_oc).astype(np.float32)
    r_co = t_co[:3, :3]
    scale = np.float32(np.linalg.det(r_co)) ** np.float32(1.0 / 3.0)
    r_co = (r_co / scale).astype(np.float32)
    r_oc = np.linalg.inv(r_co).astype(np.float32)
    ey = np.array([0., 1., 0.], np.float32)
    ng = np.array([0., -1., 0.], np.float32)
    res_rot = np.float32(1.) - np.float32(np.dot(r_co @ ey, ng))
    if res_rot < 1e-7:
        return torch.zeros(7), 0.
    j = np.zeros(7, np.float32)
    j[3:6] = np.cross(r_oc @ ng, ey)
    return torch.from_numpy(j), torch.tensor(res_rot)
