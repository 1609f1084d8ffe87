#!/usr/bin/env python3
"""Development aid: latency of small batches with the jacobian launch in the throughput (64-point tiles) and latency
(16-point tiles, rows split over waves) forms; joint optimisation and pose-only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
prm = E.gn_params()
for (B, M, Bg) in ((1, 2000, 500), (2, 2000, 500), (1, 250, 200), (4, 250, 200)):
    objs = synth.make_batch(B, first_seed=1, n_surface=M, n_background=Bg)
    b = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    for split in (0, 1, -1):
        b.set_mask_reuse(0)
        b.set_split_rows(split)
        b.run(); ts = []
        for _ in range(6):
            t0 = time.perf_counter(); b.run(); ts.append(time.perf_counter() - t0)
        st = b.stats()
        print("joint B=%d M=%d split=%2d: %.2f ms ; fwd %.2f ms (%d) jac %.2f ms (%d launches, %.3f ms each) other %.2f ms" % (
            B, M, split, np.median(ts) * 1e3, st["ms_mlp_fwd"], st["n_mlp_fwd_launches"], st["ms_mlp_jac"], st["n_mlp_jac_launches"],
            st["ms_mlp_jac"] / max(st["n_mlp_jac_launches"], 1), st["ms_total"] - st["ms_mlp_fwd"] - st["ms_mlp_jac"]), flush=True)
    b.close()
# pose-only (estimate_pose_cam_obj): 5 iterations of the surface jacobian
for B in (1, 4, 8):
    objs = synth.make_batch(B, first_seed=1, n_surface=2000, n_background=0)
    t_se3 = []
    scales = []
    for o in objs:
        t = o["t_cam_obj_init"].copy(); s = np.cbrt(np.linalg.det(t[:3, :3])); t[:3, :3] /= s
        t_se3.append(t); scales.append(s)
    codes = [np.zeros(64, np.float32)] * B
    for _ in range(2):
        eng.estimate_pose_batch(prm, t_se3, scales, [o["pts"] for o in objs], codes)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); eng.estimate_pose_batch(prm, t_se3, scales, [o["pts"] for o in objs], codes); ts.append(time.perf_counter() - t0)
    print("pose-only B=%d (auto): %.2f ms per call" % (B, np.median(ts) * 1e3), flush=True)
