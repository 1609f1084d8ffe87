"""Per-detection input preparation for the optimiser -- the numeric core of the reference's frame classes, without their
file / detector plumbing.

The reference builds `surface_points`, `rays`, `depth` and the initial `T_cam_obj` of every detection inside
`FrameWithLiDAR.get_detections` (reconstruct/kitti_sequence.py:98-216), interleaved with image / LiDAR file loading and
the Mask R-CNN / SECOND detectors.  Those stay out of scope; what the optimiser needs from them is restated here as pure
functions of arrays, in the reference's own float32 / int32 arithmetic (same expressions, same evaluation order), so the
outputs are bit-identical to the reference's (tests/test_frame_prep.py against goldens recorded from the unmodified
reference, tools/make_golden_frame.py).  Host numpy, as in the reference: at a few hundred points per detection this is
microseconds of work next to the optimiser; a GPU port is warranted only if it shows up in profiles.

New module (the reference has no such file); the arrays it returns are what `Optimizer.reconstruct_object(T_cam_obj, surface_points,
rays, depth)` takes (src/LocalMapping_util.cc:179-180).
"""
import numpy as np

from reconstruct.loss_utils import get_rays
from reconstruct.utils import ForceKeyErrorDict


def _apply_rt(points_xyz, t44):
    """Row-wise R p + t the way the reference spells it (broadcast multiply, sum over the last axis)."""
    return (points_xyz[:, None, :3] * t44[:3, :3]).sum(-1) + t44[:3, 3]


def lidar_instance(velo_pts, det_3d, t_cam_velo, max_lidar_pts):
    """One 3D detection -> instance with surface points in the camera frame and the initial Sim(3) pose.

    velo_pts (N, >=3) float32 LiDAR scan; det_3d = [x, y, z, w, l, h, theta] in the LiDAR frame (SECOND's box);
    reference kitti_sequence.py:114-157: 3 m cube pre-filter around the box centre, box test in the object frame with
    width/length enlarged by 10 %, at most `max_lidar_pts` points picked at evenly spaced ranks, T_cam_obj scaled by the
    half length."""
    det_3d = np.asarray(det_3d)
    trans, size, theta = det_3d[:3], det_3d[3:6], det_3d[6]
    c, s = np.cos(theta), np.sin(theta)
    t_velo_obj = np.array([[c, 0, -s, trans[0]],
                           [-s, 0, -c, trans[1]],
                           [0, 1, 0, trans[2] + size[2] / 2],
                           [0, 0, 0, 1]]).astype(np.float32)
    t_obj_velo = np.linalg.inv(t_velo_obj)
    reach = 3.0
    near = np.ones(velo_pts.shape[0], bool)
    for axis in range(3):
        near &= (velo_pts[:, axis] > trans[axis] - reach) & (velo_pts[:, axis] < trans[axis] + reach)
    pts_near = velo_pts[near]
    pts_obj = _apply_rt(pts_near, t_obj_velo)
    half_w, half_l, half_h = list(size / 2)
    half_w *= 1.1
    half_l *= 1.1
    limits = (half_w, half_h, half_l)                     # object frame: x = width, y = height, z = length
    inside = np.ones(pts_near.shape[0], bool)
    for axis in range(3):
        inside &= (pts_obj[:, axis] > -limits[axis]) & (pts_obj[:, axis] < limits[axis])
    pts_velo = pts_near[inside]
    n = pts_velo.shape[0]
    if n > max_lidar_pts:
        pts_velo = pts_velo[np.linspace(0, n - 1, max_lidar_pts).astype(np.int32), :]
    pts_cam = _apply_rt(pts_velo, t_cam_velo)
    t_cam_obj = t_cam_velo @ t_velo_obj
    t_cam_obj[:3, :3] *= half_l
    inst = ForceKeyErrorDict()
    inst.T_cam_obj = t_cam_obj
    inst.scale = size
    inst.surface_points = pts_cam.astype(np.float32)
    inst.num_surface_points = pts_cam.shape[0]
    inst.is_front = t_cam_obj[2, 3] > 0.0
    inst.rays = None
    return inst


def lidar_instances(velo_pts, detections_3d, t_cam_velo, max_lidar_pts):
    """All 3D detections of a frame, nearest first (kitti_sequence.py:111-113)."""
    detections_3d = np.asarray(detections_3d)
    order = np.argsort(detections_3d[:, 0])
    return [lidar_instance(velo_pts, detections_3d[i], t_cam_velo, max_lidar_pts) for i in order]


def pixels_sampler(bbox_2d, mask, downsample_ratio, img_w, img_h):
    """Grid of pixels over the 2D box (grown by 5 px, clipped to the image), every `downsample_ratio`-th, that are NOT on the
    instance mask -> (n, 2) [u, v] int32 (kitti_sequence.py:70-92 = mono_sequence.py:51-73)."""
    step = int(downsample_ratio)
    grow = 5
    last_u, last_v = img_w - 1, img_h - 1
    left, top, right, bottom = list(np.asarray(bbox_2d).astype(np.int32))
    left = left - 5 if left > grow else 0
    top = top - 5 if top > grow else 0
    right = right + 5 if right < last_u - grow else last_u
    bottom = bottom + 5 if bottom < last_v - grow else last_v
    n_rows, n_cols = bottom - top + 1, right - left + 1
    rows = np.linspace(top, bottom, int(n_rows / step)).astype(np.int32)
    cols = np.linspace(left, right, int(n_cols / step)).astype(np.int32)
    vv = np.repeat(rows, cols.shape[0])
    uu = np.tile(cols, rows.shape[0])
    off_mask = ~mask[vv, uu]
    return np.stack([uu[off_mask], vv[off_mask]], axis=-1)


def associate_masks(instances, masks_2d, bboxes_2d, k_cam, inv_k, img_w, img_h, min_mask_area, downsample_ratio,
                    max_background=200):
    """Match LiDAR instances (nearest first) with 2D instance masks and build their rays (kitti_sequence.py:177-216):
    project the surface points, pick the mask holding most of them (more than half), sample off-mask pixels of its box
    as background; rays = inv(K) [u, v, 1] for the projected surface points followed by the background pixels, depth =
    camera z of the surface points.  Also keeps the reference's running occlusion mask.  Mutates and returns `instances`."""
    if masks_2d.shape[0] == 0:
        return instances
    occluded = np.full([img_h, img_w], False, dtype=bool)
    previous = None
    for inst in instances:
        if not inst.is_front:
            continue
        pts = inst.surface_points
        uvw = (pts[:, None, :] * k_cam).sum(-1)
        uv = uvw[:, :2] / uvw[:, 2, None]
        visible = (uv[:, 0] > 0) & (uv[:, 0] < img_w) & (uv[:, 1] > 0) & (uv[:, 1] < img_h)
        px = uv[visible].astype(np.int32)
        hits = np.array([int(np.count_nonzero(masks_2d[m, px[:, 1], px[:, 0]])) for m in range(masks_2d.shape[0])])
        if hits.max() > px.shape[0] * 0.5:
            best = int(np.argmax(hits))
            inst.mask = masks_2d[best, ...]
            inst.bbox = bboxes_2d[best, ...]
            if np.count_nonzero(inst.mask) > min_mask_area:
                background = pixels_sampler(inst.bbox, inst.mask, downsample_ratio, img_w, img_h)
                if background.shape[0] > max_background:
                    keep = np.linspace(0, background.shape[0] - 1, max_background).astype(np.int32)
                    background = background[keep, :]
                inst.rays = get_rays(np.concatenate([uv, background], axis=0), inv_k).astype(np.float32)
                inst.depth = pts[:, 2].astype(np.float32)
            if previous is not None:
                occluded = occluded | previous
            inst.occ_mask = occluded
            previous = masks_2d[best, ...]
    return instances


def undistort_pixels(pixels, k_cam, k1, k2, iterations=5):
    """`cv2.undistortPoints(px, K, [k1, k2, 0, 0, 0], P=K)` for a radial two-coefficient model (mono_sequence.py:102-103).

    OpenCV is not a dependency here; this is its published fixed-point iteration (modules/calib3d undistortPoints: normalise
    with K, five iterations of  x <- x0 / (1 + k1 r^2 + k2 r^4), re-project with P), in double precision with a float32
    result like OpenCV gives for float32 input.  An iterate whose inverse distortion factor turns negative falls back to the
    normalised input point, as OpenCV does."""
    px = np.asarray(pixels, np.float32).reshape(-1, 2).astype(np.float64)
    k_cam = np.asarray(k_cam, np.float64)
    fx, fy, cx, cy = k_cam[0, 0], k_cam[1, 1], k_cam[0, 2], k_cam[1, 2]
    x0 = (px[:, 0] - cx) / fx
    y0 = (px[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    live = np.ones(x.shape[0], bool)
    for _ in range(iterations):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + (k2 * r2 + k1) * r2)
        bad = live & (icdist < 0)
        x[bad], y[bad] = x0[bad], y0[bad]
        live &= ~bad
        x = np.where(live, x0 * icdist, x)
        y = np.where(live, y0 * icdist, y)
    return np.stack([x * fx + cx, y * fy + cy], axis=-1).astype(np.float32)


def mono_instance(masks_2d, bboxes_2d, k_cam, inv_k, k1, k2, downsample_ratio, img_w, img_h, max_background=200):
    """Monocular sequences (Freiburg cars / Redwood chairs): only the detection with the largest mask is used; its box gives the
    off-mask background pixels (at most `max_background`, evenly spaced ranks), which are undistorted and turned into rays
    (mono_sequence.py:75-112).  Returns None when there is no 2D detection; surface points come later from the SLAM map
    (src/LocalMapping_util.cc:330-398), so the instance carries only `bbox`, `mask` and `background_rays`."""
    if masks_2d.shape[0] == 0:
        return None
    biggest = int(np.argmax(masks_2d.sum(axis=-1).sum(axis=-1)))
    mask = masks_2d[biggest, ...].astype(np.float32) * 255.
    bbox = bboxes_2d[biggest, ...]
    background = pixels_sampler(bbox, mask.astype(bool), downsample_ratio, img_w, img_h)
    if background.shape[0] > max_background:
        keep = np.linspace(0, background.shape[0] - 1, max_background).astype(np.int32)
        background = background[keep, :]
    undist = undistort_pixels(background, k_cam, k1, k2)
    inst = ForceKeyErrorDict()
    inst.bbox = bbox
    inst.mask = mask
    inst.background_rays = get_rays(undist, inv_k).astype(np.float32)
    return inst
