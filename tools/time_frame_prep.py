#!/usr/bin/env python3
"""Host cost of the per-detection input preparation (dsp_slam_amd/reconstruct/frame_prep.py) next to the GPU optimiser's rate:
the evidence behind keeping SURVEY 8(f) rank 3 on the host.  Uses the committed golden frame (3 detections, 6 k-point scan).

    python tools/time_frame_prep.py [--repeat 200]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsp_slam_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=200)
    ap.add_argument("--scan-points", type=int, default=0, help="pad the LiDAR scan with ground returns to this size (a KITTI sweep is ~120 k points)")
    a = ap.parse_args()
    from reconstruct import frame_prep as F
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_frame_prep.npz"))
    img_w, img_h = (int(x) for x in g["img_wh"])
    velo, boxes, t_cam_velo, masks, bboxes, k_cam, inv_k = (g[k] for k in ("velo", "boxes", "t_cam_velo", "masks", "bboxes", "k_cam", "inv_k"))
    if a.scan_points > velo.shape[0]:
        rng = np.random.default_rng(0)
        n = a.scan_points - velo.shape[0]
        pad = np.concatenate([rng.uniform(0, 80, (n, 1)), rng.uniform(-40, 40, (n, 1)), rng.normal(-1.7, 0.02, (n, 1)), rng.uniform(0, 1, (n, 1))], 1)
        velo = np.concatenate([velo, pad.astype(np.float32)], 0)
    t_lidar = t_mask = 0.0
    n_obj = 0
    for _ in range(a.repeat):
        t0 = time.perf_counter()
        insts = F.lidar_instances(velo, boxes, t_cam_velo, 250)
        t1 = time.perf_counter()
        F.associate_masks(insts, masks, bboxes, k_cam, inv_k, img_w, img_h, 1000, 4.0)
        t2 = time.perf_counter()
        t_lidar += t1 - t0
        t_mask += t2 - t1
        n_obj += len(insts)
    per_obj = (t_lidar + t_mask) / n_obj
    print("scan %d points, %d detections per frame, %d frames" % (velo.shape[0], len(boxes), a.repeat))
    print("lidar_instances %.3f ms/frame, associate_masks %.3f ms/frame -> %.3f ms per detection = %.0f detections/s on one host thread"
          % (1e3 * t_lidar / a.repeat, 1e3 * t_mask / a.repeat, 1e3 * per_obj, 1.0 / per_obj))


if __name__ == "__main__":
    main()
