#!/usr/bin/env python3
"""Development aid: step time of the bench workload (N cfg2 objects, 10 iterations) by prepass mode and pass count."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E, _lib as L
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9)
eng = E.Engine(layers, [4], 64, device=0)
prm = E.gn_params()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
objs = synth.make_batch(B, first_seed=1, n_surface=2000, n_background=500)
batch = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
combos = [(0, 0)] + [(m, p) for m in (1, 2) for p in (0, 1, 2, 3, 4, 6)]
for mode, passes in combos:
    batch.set_prepass(mode)
    batch.set_ray_passes(passes)
    batch.run()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        batch.run()
    dt = (time.perf_counter() - t0) / n
    st = batch.stats()
    other = st["ms_total"] - st["ms_mlp_fwd"] - st["ms_mlp_jac"] - st["ms_mlp_prepass"]
    print("B=%d mode=%d passes=%d: %.1f ms/step = %.1f obj/s | prepass %.1f ms (%d launches, %.3g pts, %.0f TFLOP/s) fp32 fwd %.1f ms (%d, %.3g pts, %.1f TFLOP/s) "
          "jac %.1f ms other %.1f ms | lp/insphere %.3f fp32/insphere %.3f" % (
              B, mode, passes, dt * 1e3, B / dt, st["ms_mlp_prepass"], st["n_mlp_prepass_launches"], st["n_prepass_points"],
              st["n_prepass_points"] * 3.67104e6 / max(st["ms_mlp_prepass"], 1e-9) / 1e9, st["ms_mlp_fwd"], st["n_mlp_fwd_launches"], st["n_fwd_points"],
              st["n_fwd_points"] * 3.67104e6 / max(st["ms_mlp_fwd"], 1e-9) / 1e9, st["ms_mlp_jac"], other,
              st["n_prepass_points"] / st["n_insphere_points"], st["n_fwd_points"] / st["n_insphere_points"]), flush=True)
