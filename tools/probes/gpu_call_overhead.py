#!/usr/bin/env python3
"""Development aid: per-call cost of the drop-in single-object entry points (allocation + upload + run + download)
versus a resident batch re-run."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9)
eng = E.Engine(layers, [4], 64, device=0)
prm = E.gn_params()
for (M, Bg, name) in ((2000, 500, "cfg2"), (250, 200, "kitti-real-size")):
    o = synth.make_object(1, n_surface=M, n_background=Bg)
    args = ([o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
    eng.reconstruct_batch(prm, *args)
    t = []
    for _ in range(7):
        t0 = time.perf_counter(); eng.reconstruct_batch(prm, *args); t.append((time.perf_counter() - t0) * 1e3)
    b = eng.batch(prm, *args)
    b.run()
    r = []
    for _ in range(7):
        t0 = time.perf_counter(); b.run(); b.results(); r.append((time.perf_counter() - t0) * 1e3)
    b.close()
    print("%s: one-shot call median %.2f ms, resident re-run median %.2f ms -> per-call overhead %.2f ms" % (name, np.median(t), np.median(r), np.median(t) - np.median(r)))
    pose = []
    s = float(o["scale"]); tt = o["t_cam_obj_init"].copy(); tt[:3, :3] /= s
    code = np.zeros(64, np.float32)
    eng.estimate_pose_batch(prm, [tt], [s], [o["pts"]], [code])
    for _ in range(7):
        t0 = time.perf_counter(); eng.estimate_pose_batch(prm, [tt], [s], [o["pts"]], [code]); pose.append((time.perf_counter() - t0) * 1e3)
    print("   estimate_pose_cam_obj one-shot median %.2f ms" % np.median(pose))
