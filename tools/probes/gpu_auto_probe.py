#!/usr/bin/env python3
"""Development aid: throughput of the automatic (adaptive) ray-pass mode for several batch sizes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
prm = E.gn_params()
for B in (32, 8, 4, 1):
    objs = synth.make_batch(B, first_seed=1, n_surface=2000, n_background=500)
    b = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    for mode in (0, 10 if B >= 8 else 2):
        b.set_ray_passes(mode)
        b.run(); ts = []
        for _ in range(3):
            t0 = time.perf_counter(); b.run(); ts.append(time.perf_counter() - t0)
        st = b.stats()
        print("B=%d passes=%s: %.1f ms -> %.2f obj/s ; evaluated %.1f%% ; fwd launches %d, fwd ms %.1f" % (
            B, "auto" if mode == 0 else mode, np.median(ts) * 1e3, B / np.median(ts), 100 * st["n_fwd_points"] / st["n_insphere_points"],
            st["n_mlp_fwd_launches"], st["ms_mlp_fwd"]), flush=True)
    b.close()
