#!/usr/bin/env python3
"""tests/golden/golden_lie.npz: the reference's exp_sim3 / exp_se3 / compute_rotation_loss_sim3 and its Sim(3) state update on a set of
arguments that reaches every branch, recorded by running the UNMODIFIED reference (oracle/ref_shim.py) on the CPU.

    python tools/make_golden_lie.py

Covers reconstruct/loss_utils.py:129-163 (exp_se3: theta <= 1e-8 and the general branch), :188-233 (exp_sim3: the theta <= 1e-8 branch
with s == 0 and s != 0, the `c = 0. if s <= eps` quirk at s < 0, s == 0, s == 1e-8 exactly and just above it, rotations near pi),
reconstruct/loss.py:155-178 (rotation prior: the res < 1e-7 zero branch, tilted, scaled) and reconstruct/optimizer.py:120-125,187-188
(scale, depth range, `exp_sim3(lr dx) @ t_obj_cam`).  Consumed by tests/test_gpu_lie.py (device) and tests/test_oracle_golden.py (oracle).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

GOLD = os.environ.get("DSP_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))      # (tests regenerate into a scratch directory)


def main():
    ref_shim.install()
    import reconstruct.loss as rloss
    import reconstruct.loss_utils as rlu
    torch.set_num_threads(1)
    rng = np.random.default_rng(20260926)
    f = np.float32

    xs = [
        # the seven vectors of golden_terms.npz (kept so that both files agree)
        [0.1, -0.2, 0.3, 0.02, -0.01, 0.03, 0.05], [0.1, -0.2, 0.3, 0.02, -0.01, 0.03, -0.05],
        [0.1, -0.2, 0.3, 0.02, -0.01, 0.03, 0.0], [0.1, -0.2, 0.3, 0.0, 0.0, 0.0, 0.04],
        [0.1, -0.2, 0.3, 0.0, 0.0, 0.0, 0.0], [0.5, 0.1, -0.7, 1.2, -0.4, 0.8, 0.3],
        [-0.01, 0.004, 0.02, 1e-4, -2e-4, 5e-5, 1e-3],
        # theta on either side of the 1e-8 branch (theta = |w|)
        [0.3, 0.2, -0.1, 1e-8, 0.0, 0.0, 0.02], [0.3, 0.2, -0.1, 2e-8, 0.0, 0.0, 0.02], [0.3, 0.2, -0.1, 6e-9, 6e-9, 0.0, -0.02],
        [0.3, 0.2, -0.1, 0.0, 1e-6, 0.0, 0.0], [0.3, 0.2, -0.1, 1e-5, 1e-5, -1e-5, 1e-9],
        # s on either side of the `s <= eps` quirk, theta > 0
        [0.3, 0.2, -0.1, 0.02, 0.01, -0.03, 1e-8], [0.3, 0.2, -0.1, 0.02, 0.01, -0.03, 2e-8], [0.3, 0.2, -0.1, 0.02, 0.01, -0.03, -1e-9],
        [0.3, 0.2, -0.1, 0.02, 0.01, -0.03, 1e-6], [0.3, 0.2, -0.1, 0.02, 0.01, -0.03, -0.3],
        # theta <= 1e-8 with s < 0 (the first branch has no quirk: c = (e^s - 1) / s)
        [0.3, 0.2, -0.1, 0.0, 0.0, 0.0, -0.07],
        # large rotations, near pi
        [0.2, -0.4, 0.1, 3.1, 0.0, 0.0, 0.1], [0.2, -0.4, 0.1, 1.8, -1.8, 1.8, -0.1], [0.2, -0.4, 0.1, 0.0, 3.14159, 0.0, 0.2],
        # a typical first Gauss-Newton step of the bench objects (25 cm, 5 degrees, a few % of scale), and a typical last one
        [0.11, -0.02, -0.19, 0.004, 0.08, -0.003, 0.012], [2e-4, -1e-4, 3e-4, 1e-5, 2e-4, -1e-5, -3e-5],
    ]
    for _ in range(9):
        xs.append(np.concatenate([rng.normal(size=3) * 0.3, rng.normal(size=3) * rng.choice([1e-3, 0.05, 0.7]), [rng.normal() * 0.1]]))
    xs = [np.array(v, f) for v in xs]
    out = {"exp_x": np.stack(xs)}
    out["exp_sim3"] = np.stack([rlu.exp_sim3(torch.from_numpy(v.copy())).numpy() for v in xs])
    out["exp_se3"] = np.stack([rlu.exp_se3(torch.from_numpy(v[:6].copy())).numpy() for v in xs])

    # rotation prior + derived state: an upright object (y axis = camera -y: the zero branch), tilts of 1e-4 .. 0.7 rad about several axes,
    # scales 0.5 .. 3, 8-25 m ahead -- t_obj_cam = inv(t_cam_obj) computed by torch.inverse as the reference's callers do
    def t_cam_obj(scale, yaw, tilt_axis, tilt, t):
        c, s = np.cos(yaw), np.sin(yaw)
        ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
        ax = np.asarray(tilt_axis, np.float64)
        ax = ax / np.linalg.norm(ax)
        k = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        rt = np.eye(3) + np.sin(tilt) * k + (1 - np.cos(tilt)) * k @ k
        m = np.eye(4)
        m[:3, :3] = scale * rt @ ry @ np.diag([1.0, -1.0, -1.0])
        m[:3, 3] = t
        return m.astype(f)

    poses = [t_cam_obj(2.0, 0.3, (1, 0, 0), 0.0, (1.0, 1.2, 12.0)), t_cam_obj(1.0, -2.0, (1, 0, 0), 0.0, (-3.0, 1.2, 20.0)),
             t_cam_obj(2.0, 0.3, (1, 0, 0), 1e-4, (1.0, 1.2, 12.0)), t_cam_obj(2.0, 0.3, (0, 0, 1), 3e-4, (1.0, 1.2, 12.0)),
             t_cam_obj(1.8, 1.0, (1, 0, 1), 0.05, (0.5, 1.0, 9.0)), t_cam_obj(2.2, -0.7, (0.3, 0.2, 0.9), 0.7, (-2.0, 1.4, 25.0)),
             t_cam_obj(0.5, 2.5, (1, 0, 0), 0.2, (0.0, 0.3, 3.0)), t_cam_obj(3.0, 0.0, (0, 0, 1), -0.4, (4.0, 1.2, 8.0))]
    rot_t, rot_j, rot_r, rot_scale, rot_range = [], [], [], [], []
    for p in poses:
        t_oc = torch.inverse(torch.from_numpy(p))
        jr, rres = rloss.compute_rotation_loss_sim3(t_oc.clone())
        # optimizer.py:120-125
        t_co = torch.inverse(t_oc)
        scale = torch.det(t_co[:3, :3]) ** (1 / 3)
        dmin, dmax = t_co[2, 3] - 1.0 * scale, t_co[2, 3] + 1.0 * scale
        rot_t.append(t_oc.numpy().copy())
        rot_j.append(jr.numpy().copy())
        rot_r.append(f(rres))
        rot_scale.append(f(scale))
        rot_range.append(np.array([f(dmin), f(dmax)], f))
    out.update(rot_t=np.stack(rot_t), rot_j=np.stack(rot_j), rot_r=np.array(rot_r, f), rot_scale=np.array(rot_scale, f),
               rot_range=np.stack(rot_range))
    # upright and the 1e-4 rad tilt (res = 5e-9 -> 0 in float32) take the zero branch; the 3e-4 rad tilt is ONE float32 step above it (1.19e-7)
    assert (out["rot_r"][:3] == 0.0).all() and (out["rot_r"][3:] > 1e-7).all() and out["rot_r"][3] < 2e-7, out["rot_r"]

    # the state update of optimizer.py:187-188: t_obj_cam <- exp_sim3(lr * dx[:7]) @ t_obj_cam (lr = 1)
    upd_t, upd_dx, upd_out = [], [], []
    for i, p in enumerate(poses):
        t_oc = torch.inverse(torch.from_numpy(p))
        for dx in (xs[21], xs[22], xs[1], xs[5]):
            upd_t.append(t_oc.numpy().copy())
            upd_dx.append(dx)
            upd_out.append(torch.mm(rlu.exp_sim3(1.0 * torch.from_numpy(dx.copy())), t_oc).numpy())
    out.update(upd_t=np.stack(upd_t), upd_dx=np.stack(upd_dx), upd_out=np.stack(upd_out))
    np.savez_compressed(os.path.join(GOLD, "golden_lie.npz"), **out)
    print("golden_lie.npz: %d exp vectors, %d rotation-prior poses (res: %s), %d updates" % (
        len(xs), len(poses), " ".join("%.3g" % r for r in out["rot_r"]), len(upd_t)))


if __name__ == "__main__":
    main()
