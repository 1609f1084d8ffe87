#!/usr/bin/env python3
"""Development aid: throughput / latency versus the number of front-to-back ray passes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9)
eng = E.Engine(layers, [4], 64, device=0)
prm = E.gn_params()
SWEEPS = ((32, (1, 3, 5, 7, 10, 13, 17, 25)), (1, (1, 2, 3, 4, 5, 7, 10)), (4, (2, 3, 5, 7, 10)))
if len(sys.argv) > 2:      # python tools/probes/gpu_pass_sweep.py <objects> <passes,passes,...>
    SWEEPS = ((int(sys.argv[1]), tuple(int(x) for x in sys.argv[2].split(","))),)
for B, sweep in SWEEPS:
    objs = synth.make_batch(B, first_seed=1, n_surface=2000, n_background=500)
    b = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    for P in sweep:
        b.set_ray_passes(P)
        b.run()
        ts = []
        for _ in range(2 if B > 8 else 5):
            t0 = time.perf_counter(); b.run(); ts.append(time.perf_counter() - t0)
        st = b.stats()
        print("B=%d passes=%2d: %.1f ms/run -> %.2f obj/s ; evaluated %.0f%% of V ; fwd kernel %.1f ms" % (
            B, P, np.median(ts) * 1e3, B / np.median(ts), 100 * st["n_fwd_points"] / st["n_insphere_points"], st["ms_mlp_fwd"]), flush=True)
    b.close()
