#!/usr/bin/env python3
"""Golden vectors for the frame-preparation mirror (dsp_slam_amd/reconstruct/frame_prep.py), recorded from the UNMODIFIED
reference: FrameWithLiDAR.get_detections / pixels_sampler (reconstruct/kitti_sequence.py:70-216) run on a synthetic frame.

Runs only where /root/reference exists (the build container).  The reference class is instantiated without its __init__
(which loads image / LiDAR files through cv2) and handed arrays plus stand-in detectors; `cv2` is stubbed because it is not
installed here and is not touched by the methods used, and `np.bool` (removed from numpy >= 1.24, used at
kitti_sequence.py:181) is aliased to `bool` for the duration of the run.

    python tools/make_golden_frame.py        ->  tests/golden/golden_frame_prep.npz, golden_mono_prep.npz

The monocular frame class (reconstruct/mono_sequence.py:51-112) is recorded the same way.  Its one OpenCV call,
`cv2.undistortPoints`, cannot run here: the stub records the pixels it was handed and returns them unchanged, which is what
OpenCV returns for the zero distortion the golden is recorded with (to float32 rounding of integer pixel coordinates: exactly).
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def synthetic_frame(seed=0):
    """A KITTI-like frame: a LiDAR scan with three car-sized point clusters on a ground plane, their 3D boxes, instance
    masks around their image projections, plus one mask that matches nothing."""
    rng = np.random.default_rng(seed)
    img_w, img_h = 1226, 370
    k_cam = np.array([[707.0912, 0.0, 601.8873], [0.0, 707.0912, 183.1104], [0.0, 0.0, 1.0]])
    inv_k = np.linalg.inv(k_cam)
    # KITTI velo -> cam: x_cam = -y_velo, y_cam = -z_velo, z_cam = x_velo (+ small offsets)
    t_cam_velo = np.array([[0.0, -1.0, 0.0, 0.0], [0.0, 0.0, -1.0, -0.08], [1.0, 0.0, 0.0, -0.27], [0.0, 0.0, 0.0, 1.0]], dtype=np.float32)
    boxes = np.array([[14.0, 2.5, -1.0, 1.7, 4.2, 1.5, 0.3],
                      [8.0, -3.0, -0.9, 1.6, 3.9, 1.45, -1.2],
                      [25.0, 0.5, -1.1, 1.8, 4.5, 1.6, 1.6]], dtype=np.float32)
    pts = [np.concatenate([rng.uniform(0, 40, (6000, 1)), rng.uniform(-15, 15, (6000, 1)), rng.normal(-1.7, 0.02, (6000, 1))], 1)]
    for (x, y, z, w, l, h, th) in boxes:
        n = int(rng.integers(180, 900))
        local = np.stack([rng.uniform(-w / 2, w / 2, n), rng.uniform(0, h, n), rng.uniform(-l / 2, l / 2, n)], 1)
        face = rng.integers(0, 3, n)                      # points on box faces, like a LiDAR return
        local[face == 0, 0] = np.sign(local[face == 0, 0]) * w / 2
        local[face == 1, 2] = np.sign(local[face == 1, 2]) * l / 2
        c, s = np.cos(th), np.sin(th)
        t_velo_obj = np.array([[c, 0, -s, x], [-s, 0, -c, y], [0, 1, 0, z + h / 2 - h / 2]])   # object y up from the box bottom
        pts.append(local @ t_velo_obj[:, :3].T + t_velo_obj[:, 3])
    velo = np.concatenate(pts, 0)
    velo = np.concatenate([velo, rng.uniform(0, 1, (velo.shape[0], 1))], 1).astype(np.float32)
    velo = velo[rng.permutation(velo.shape[0])]
    # masks: filled ellipses around the projected clusters
    masks, bboxes = [], []
    vv, uu = np.mgrid[0:img_h, 0:img_w]
    for (x, y, z, w, l, h, th) in boxes:
        centre_cam = t_cam_velo[:3, :3] @ np.array([x, y, z + h / 2]) + t_cam_velo[:3, 3]
        u0, v0 = (k_cam @ centre_cam)[:2] / centre_cam[2]
        ru, rv = 707.0 * 0.5 * max(l, w) / centre_cam[2], 707.0 * 0.5 * h / centre_cam[2]
        m = ((uu - u0) / (ru * 1.1)) ** 2 + ((vv - v0) / (rv * 1.3)) ** 2 < 1.0
        masks.append(m)
        bboxes.append([u0 - ru, v0 - rv, u0 + ru, v0 + rv])
    masks.append(((uu - 100) / 30.0) ** 2 + ((vv - 60) / 20.0) ** 2 < 1.0)
    bboxes.append([70, 40, 130, 80])
    return dict(img_w=img_w, img_h=img_h, k_cam=k_cam, inv_k=inv_k, t_cam_velo=t_cam_velo, boxes=boxes, velo=velo,
                masks=np.stack(masks), bboxes=np.array(bboxes, dtype=np.float32))


def record_mono(fr):
    """reconstruct/mono_sequence.py Frame.get_detections on the same masks (Redwood-like: the largest mask wins)."""
    import cv2
    from reconstruct.mono_sequence import Frame
    handed = {}

    def undistort_stub(pts, k, dist, P=None):
        handed["pts"], handed["dist"] = np.array(pts), np.array(dist)
        return np.array(pts)

    cv2.undistortPoints = undistort_stub
    if not hasattr(np, "bool8"):
        np.bool8 = np.bool_                              # mono_sequence.py:96 predates numpy 2
    frame = Frame.__new__(Frame)
    frame.configs = types.SimpleNamespace(downsample_ratio=4.0)
    frame.K, frame.invK, frame.k1, frame.k2 = fr["k_cam"], fr["inv_k"], 0.0, 0.0
    frame.online = True
    frame.object_class = "chairs"
    frame.img_rgb = np.zeros((fr["img_h"], fr["img_w"], 3), np.uint8)
    frame.img_bgr = frame.img_rgb
    frame.img_h, frame.img_w = fr["img_h"], fr["img_w"]
    frame.instances = []
    frame.detector_2d = types.SimpleNamespace(make_prediction=lambda img, object_class=None: {"pred_masks": fr["masks"], "pred_boxes": fr["bboxes"]})
    frame.get_detections()
    inst = frame.instances[0]
    out = dict(masks=fr["masks"], bboxes=fr["bboxes"], k_cam=fr["k_cam"], inv_k=fr["inv_k"], img_wh=np.array([fr["img_w"], fr["img_h"]]),
               bbox=inst.bbox, mask=inst.mask, background_rays=inst.background_rays, handed_pixels=handed["pts"], handed_dist=handed["dist"])
    np.savez_compressed(os.path.join(GOLD, "golden_mono_prep.npz"), **out)
    print("mono: background rays", inst.background_rays.shape, inst.background_rays.dtype, "pixels handed to cv2", handed["pts"].shape, handed["pts"].dtype)


def main():
    from oracle import ref_shim
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    ref_shim.install()
    had_bool = hasattr(np, "bool")
    if not had_bool:
        np.bool = bool                                   # kitti_sequence.py:181 predates numpy 1.24
    try:
        import torch
        from reconstruct.kitti_sequence import FrameWithLiDAR
        fr = synthetic_frame(0)
        frame = FrameWithLiDAR.__new__(FrameWithLiDAR)
        cfg = types.SimpleNamespace(downsample_ratio=4.0)
        frame.configs = cfg
        frame.K, frame.invK, frame.T_cam_velo = fr["k_cam"], fr["inv_k"], fr["t_cam_velo"]
        frame.online = True
        frame.max_lidar_pts, frame.min_lidar_pts, frame.min_mask_area = 250, 10, 1000
        frame.velo_pts = fr["velo"]
        frame.velo_file = None
        frame.img_rgb = np.zeros((fr["img_h"], fr["img_w"], 3), np.uint8)
        frame.img_bgr = frame.img_rgb
        frame.img_h, frame.img_w = fr["img_h"], fr["img_w"]
        frame.instances = []
        frame.detector_3d = types.SimpleNamespace(make_prediction=lambda f: torch.from_numpy(fr["boxes"]))
        frame.detector_2d = types.SimpleNamespace(make_prediction=lambda img: {"pred_masks": fr["masks"], "pred_boxes": fr["bboxes"]})
        frame.get_detections()
        out = {k: fr[k] for k in ("k_cam", "inv_k", "t_cam_velo", "boxes", "velo", "masks", "bboxes")}
        out["img_wh"] = np.array([fr["img_w"], fr["img_h"]])
        out["n_instances"] = np.array(len(frame.instances))
        for i, inst in enumerate(frame.instances):
            out["i%d_T_cam_obj" % i] = inst.T_cam_obj
            out["i%d_surface_points" % i] = inst.surface_points
            out["i%d_is_front" % i] = np.array(bool(inst.is_front))
            out["i%d_has_rays" % i] = np.array(inst.rays is not None)
            if inst.rays is not None:
                out["i%d_rays" % i] = inst.rays
                out["i%d_depth" % i] = inst.depth
                out["i%d_bbox" % i] = inst.bbox
                out["i%d_occ_sum" % i] = np.array(int(inst.occ_mask.sum()))
        # the sampler alone, on a box touching the image border
        out["sampler_bbox"] = np.array([1200.4, 300.2, 1225.9, 369.0], np.float32)
        out["sampler_out"] = frame.pixels_sampler(out["sampler_bbox"], fr["masks"][0])
        os.makedirs(GOLD, exist_ok=True)
        np.savez_compressed(os.path.join(GOLD, "golden_frame_prep.npz"), **out)
        print("instances:", len(frame.instances), [(int(i.num_surface_points), None if i.rays is None else i.rays.shape) for i in frame.instances])
        record_mono(fr)
    finally:
        if not had_bool:
            del np.bool


if __name__ == "__main__":
    main()
