"""Decoder queries and small Lie-group / robust-kernel helpers -- mirror of reference reconstruct/loss_utils.py.

decode_sdf / get_batch_sdf_jacobian run on the GPU (HIP kernels K1/K2).  The remaining functions are
O(1)-sized host helpers kept for API parity; inside Optimizer their device counterparts are used
(dsp_slam_amd/csrc/gn_kernels.hip), so no iteration of the optimiser round-trips through them.
Tensors are returned as CPU torch tensors with the reference's shapes.
"""
import time

import numpy as np
import torch

F32 = np.float32


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def get_rays(sampled_pixels, invK):
    """Pixel coordinates (N, 2) as [u, v] and the inverse intrinsics (3, 3) -> ray directions (N, 3) in the camera frame, float32
    (reference loss_utils.py:23-37: invK @ [u, v, 1] in float64, terms added left to right, rounded once)."""
    px = np.asarray(sampled_pixels, dtype=np.result_type(np.asarray(sampled_pixels).dtype, np.float64))
    k = np.asarray(invK)
    return ((px[:, 0:1] * k[:, 0] + px[:, 1:2] * k[:, 1]) + k[:, 2]).astype(np.float32)


def sdf_to_occupancy(sdf_tensor, th=0.015):
    """reference loss_utils.py:40-48."""
    return 0.5 - torch.clamp(sdf_tensor, min=-th, max=th) / (2 * th)


def decode_sdf(decoder, lat_vec, x, max_batch=64 ** 3):
    """reference loss_utils.py:51-79: (N,3) -> (N,) sdf, forward only, on the GPU."""
    return torch.from_numpy(decoder.engine.decode_sdf(_np(lat_vec), _np(x)[:, 0:3]))


def get_batch_sdf_jacobian(decoder, lat_vec, x, out_dim=1):
    """reference loss_utils.py:82-103 -> y (N,1,1), d y / d [code, xyz] (N,1,code_len+3)."""
    if out_dim != 1:
        raise NotImplementedError("out_dim != 1")
    sdf, grad = decoder.engine.sdf_jacobian(_np(lat_vec), _np(x))
    return torch.from_numpy(sdf).view(-1, 1, 1), torch.from_numpy(grad).view(grad.shape[0], 1, -1)


def get_points_to_pose_jacobian_se3(points):
    """reference loss_utils.py:107-126: [I | -[p]x] (N,3,6)."""
    p = _np(points).astype(F32)
    j = np.zeros((p.shape[0], 3, 6), F32)
    j[:, 0, 0] = j[:, 1, 1] = j[:, 2, 2] = 1
    j[:, 0, 4], j[:, 0, 5] = p[:, 2], -p[:, 1]
    j[:, 1, 3], j[:, 1, 5] = -p[:, 2], p[:, 0]
    j[:, 2, 3], j[:, 2, 4] = p[:, 1], -p[:, 0]
    return torch.from_numpy(j)


def get_points_to_pose_jacobian_sim3(points):
    """reference loss_utils.py:166-185: [I | -[p]x | p] (N,3,7)."""
    p = torch.from_numpy(_np(points).astype(F32))
    return torch.cat((get_points_to_pose_jacobian_se3(p), p[..., None]), dim=-1)


def _so3_terms(w):
    w_hat = np.array([[0., -w[2], w[1]], [w[2], 0., -w[0]], [-w[1], w[0], 0.]], F32)
    return w_hat, (w_hat @ w_hat).astype(F32), F32(np.sqrt(np.sum(w * w, dtype=F32)))


def exp_se3(x):
    """reference loss_utils.py:129-163."""
    x = _np(x).astype(F32)
    w_hat, w_hat2, theta = _so3_terms(x[3:6])
    eye = np.eye(3, dtype=F32)
    if theta <= 1e-8:
        e_w, j = eye, eye
    else:
        s, c = F32(np.sin(np.float64(theta))), F32(np.cos(np.float64(theta)))     # correctly rounded, as the reference's torch returns them (95 %)
        e_w = eye + w_hat * s / theta + w_hat2 * (F32(1) - c) / theta ** 2
        j = eye + ((F32(1) - c) / theta ** 2) * w_hat + ((theta - s) / theta ** 3) * w_hat2
    out = np.eye(4, dtype=F32)
    out[:3, :3] = e_w
    out[:3, 3] = j.astype(F32) @ x[:3]
    return torch.from_numpy(out)


def exp_sim3(x):
    """reference loss_utils.py:188-233, including its `c = 0 if s <= eps` branch (:223)."""
    x = _np(x).astype(F32)
    w_hat, w_hat2, theta = _so3_terms(x[3:6])
    s = F32(x[6])
    e_s = F32(np.exp(np.float64(s)))       # correctly rounded like torch's (numpy's float32 exp is one ulp off for 40 % of arguments, and (e^s - 1) / s amplifies that by 1 / s)
    eye = np.eye(3, dtype=F32)
    if theta <= 1e-8:
        e_w = eye
        j = eye if s == 0 else ((e_s - F32(1)) / s) * eye
    else:
        sn, cs = F32(np.sin(np.float64(theta))), F32(np.cos(np.float64(theta)))
        t2, s2 = F32(theta ** 2), F32(s ** 2)
        e_w = eye + w_hat * sn / theta + w_hat2 * (F32(1) - cs) / t2
        a, b = e_s * sn, e_s * cs
        c = F32(0) if s <= 1e-8 else (e_s - F32(1)) / s
        k1 = (a * s + (F32(1) - b) * theta) / (s2 + t2)
        k2 = c - ((b - F32(1)) * s + a * theta) / (s2 + t2)
        j = c * eye + k1 * w_hat / theta + k2 * w_hat2 / t2
    out = np.eye(4, dtype=F32)
    out[:3, :3] = e_s * e_w
    out[:3, 3] = j.astype(F32) @ x[:3]
    return torch.from_numpy(out)


def huber_norm_weights(x, b=0.02):
    """reference loss_utils.py:236-248 (x = residual norms)."""
    xn = _np(x).astype(F32).copy()
    rho = np.where(xn <= b, xn * xn, F32(2 * b) * xn - F32(b * b)).astype(F32)
    xn[xn == 0] = 1.
    return torch.from_numpy((np.sqrt(rho) / xn).astype(F32)).view(x.shape if hasattr(x, "shape") else -1)


def get_robust_res(res, b):
    """reference loss_utils.py:251-265."""
    res = torch.as_tensor(res).view(-1, 1, 1)
    w = huber_norm_weights(torch.abs(res), b=b)
    robust_res = w * res
    return robust_res, torch.mean(robust_res ** 2), w


def get_time():
    """reference loss_utils.py:268-273 (every engine call is synchronous, nothing to wait for)."""
    return time.time()
