"""CPU-only: the frame-preparation mirror (dsp_slam_amd/reconstruct/frame_prep.py) against goldens recorded
from the UNMODIFIED reference (FrameWithLiDAR.get_detections / pixels_sampler, reconstruct/kitti_sequence.py:70-216, run by
tools/make_golden_frame.py on a synthetic KITTI-like frame).  Same float32 / int32 arithmetic, so the comparison is bit for bit."""
import os
import sys

import numpy as np
import pytest

from conftest import golden

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsp_slam_amd")


@pytest.fixture
def mirror():
    sys.path.insert(0, PKG)
    for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
        del sys.modules[m]
    yield
    sys.path.remove(PKG)
    for m in [k for k in sys.modules if k.split(".")[0] in ("reconstruct", "deep_sdf")]:
        del sys.modules[m]


def test_lidar_instances_and_rays_equal_the_reference(mirror):
    from reconstruct import frame_prep as F
    g = golden("golden_frame_prep.npz")
    img_w, img_h = (int(x) for x in g["img_wh"])
    insts = F.lidar_instances(g["velo"], g["boxes"], g["t_cam_velo"], max_lidar_pts=250)
    assert len(insts) == int(g["n_instances"]) == 3
    for i, inst in enumerate(insts):
        assert inst.surface_points.dtype == np.float32
        assert np.array_equal(inst.surface_points, g["i%d_surface_points" % i])
        assert np.array_equal(inst.T_cam_obj, g["i%d_T_cam_obj" % i]) and inst.T_cam_obj.dtype == g["i%d_T_cam_obj" % i].dtype
        assert bool(inst.is_front) == bool(g["i%d_is_front" % i])
        assert inst.num_surface_points <= 250
    F.associate_masks(insts, g["masks"], g["bboxes"], g["k_cam"], g["inv_k"], img_w, img_h, min_mask_area=1000, downsample_ratio=4.0)
    n_with_rays = 0
    for i, inst in enumerate(insts):
        has = inst.rays is not None
        assert has == bool(g["i%d_has_rays" % i])
        if has:
            n_with_rays += 1
            assert inst.rays.dtype == np.float32 and np.array_equal(inst.rays, g["i%d_rays" % i])
            assert np.array_equal(inst.depth, g["i%d_depth" % i])
            assert np.array_equal(inst.bbox, g["i%d_bbox" % i])
            assert int(inst.occ_mask.sum()) == int(g["i%d_occ_sum" % i])
            # the optimiser's calling convention: the first M rays belong to the M surface points (depth per ray), the rest are background
            assert inst.rays.shape[0] >= inst.depth.shape[0] == inst.surface_points.shape[0]
    assert n_with_rays == 3


def test_pixels_sampler_border_box(mirror):
    from reconstruct import frame_prep as F
    g = golden("golden_frame_prep.npz")
    img_w, img_h = (int(x) for x in g["img_wh"])
    out = F.pixels_sampler(g["sampler_bbox"], g["masks"][0], 4.0, img_w, img_h)
    assert out.dtype == g["sampler_out"].dtype and np.array_equal(out, g["sampler_out"])
    assert out[:, 0].max() <= img_w - 1 and out[:, 1].max() <= img_h - 1


def test_no_masks_and_far_detections(mirror):
    from reconstruct import frame_prep as F
    g = golden("golden_frame_prep.npz")
    insts = F.lidar_instances(g["velo"], g["boxes"][:1], g["t_cam_velo"], max_lidar_pts=250)
    F.associate_masks(insts, g["masks"][:0], g["bboxes"][:0], g["k_cam"], g["inv_k"], 1226, 370, 1000, 4.0)
    assert insts[0].rays is None
    # a box with no LiDAR return inside: zero surface points, still a valid instance
    empty = F.lidar_instance(g["velo"], np.array([200.0, 0.0, -1.0, 1.7, 4.2, 1.5, 0.0], np.float32), g["t_cam_velo"], 250)
    assert empty.num_surface_points == 0 and empty.surface_points.shape == (0, 3)


def test_mono_instance_equals_the_reference(mirror):
    """mono_sequence.py:75-112 (largest mask, off-mask pixels of its box, at most 200, undistort, rays), recorded from the
    unmodified reference with zero distortion (cv2 stubbed as pass-through: tools/make_golden_frame.py)."""
    from reconstruct import frame_prep as F
    g = golden("golden_mono_prep.npz")
    img_w, img_h = (int(x) for x in g["img_wh"])
    inst = F.mono_instance(g["masks"], g["bboxes"], g["k_cam"], g["inv_k"], 0.0, 0.0, 4.0, img_w, img_h)
    assert np.array_equal(inst.bbox, g["bbox"]) and np.array_equal(inst.mask, g["mask"]) and inst.mask.dtype == g["mask"].dtype
    assert inst.background_rays.dtype == np.float32 and np.array_equal(inst.background_rays, g["background_rays"])
    assert g["handed_dist"].tolist() == [0.0, 0.0, 0.0, 0.0, 0.0]
    # zero distortion: the undistorted pixels are the pixels, bit for bit (what OpenCV returns for integer pixel coordinates)
    px = g["handed_pixels"].reshape(-1, 2)
    assert np.array_equal(F.undistort_pixels(px, g["k_cam"], 0.0, 0.0), px)
    assert F.mono_instance(g["masks"][:0], g["bboxes"][:0], g["k_cam"], g["inv_k"], 0.0, 0.0, 4.0, img_w, img_h) is None


def test_undistort_inverts_the_radial_model(mirror):
    """OpenCV's fixed-point iteration (cv2 is absent: parity with it is unpinned): re-applying the forward model
    x_d = x (1 + k1 r^2 + k2 r^4) to the result reproduces the input pixels; Freiburg-like coefficients."""
    from reconstruct import frame_prep as F
    k_cam = np.array([[535.4, 0.0, 320.1], [0.0, 539.2, 247.6], [0.0, 0.0, 1.0]])
    rng = np.random.default_rng(0)
    px = np.stack([rng.uniform(0, 640, 500), rng.uniform(0, 480, 500)], 1).astype(np.float32)
    def redistort(und, k1, k2):
        und = und.astype(np.float64)
        x, y = (und[:, 0] - k_cam[0, 2]) / k_cam[0, 0], (und[:, 1] - k_cam[1, 2]) / k_cam[1, 1]
        r2 = x * x + y * y
        f = 1 + k1 * r2 + k2 * r2 * r2
        return np.stack([x * f * k_cam[0, 0] + k_cam[0, 2], y * f * k_cam[1, 1] + k_cam[1, 2]], 1)

    centre = (np.abs(px[:, 0] - 320) < 160) & (np.abs(px[:, 1] - 240) < 120)
    for k1, k2 in ((0.0, 0.0), (0.2312, -0.7849), (-0.28, 0.07)):
        und = F.undistort_pixels(px, k_cam, k1, k2)
        assert und.dtype == np.float32
        # five iterations (OpenCV's default) are tight in the middle of the image, looser towards the corners ...
        assert np.abs(redistort(und, k1, k2) - px)[centre].max() < 0.02
        # ... and the iteration converges to the inverse of the model
        # (the strongly non-monotonic k2 of the second set has no inverse in the far corners: only the middle is checked there)
        sel = centre if k2 < -0.5 else np.ones_like(centre)
        assert np.abs(redistort(F.undistort_pixels(px, k_cam, k1, k2, iterations=60), k1, k2) - px)[sel].max() < 2e-3
        if k1 != 0:
            assert np.abs(und - px).max() > 1.0
