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
