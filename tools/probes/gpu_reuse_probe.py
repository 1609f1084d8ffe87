#!/usr/bin/env python3
"""Development aid: throughput / latency with the render rows' backward-only launch (mask reuse) off, on and automatic."""
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
    for mode in (0, 1, -1):
        b.set_mask_reuse(mode)
        b.run(); ts = []
        for _ in range(4):
            t0 = time.perf_counter(); b.run(); ts.append(time.perf_counter() - t0)
        st = b.stats()
        ff = st["n_fwd_points"] * 3.67104e6 / (st["ms_mlp_fwd"] * 1e-3) / 1e12
        jf = (st["n_jac_points"] * 7.34208e6 + st["n_render_rows"] * 3.67104e6) / (st["ms_mlp_jac"] * 1e-3) / 1e12
        print("B=%d reuse=%d: %.2f ms -> %.2f obj/s ; fwd %.1f ms (%.1f TFLOP/s, %d launches) jac %.1f ms (%.1f TFLOP/s, %d launches) other %.1f ms" % (
            B, mode, np.median(ts) * 1e3, B / np.median(ts), st["ms_mlp_fwd"], ff, st["n_mlp_fwd_launches"], st["ms_mlp_jac"], jf,
            st["n_mlp_jac_launches"], st["ms_total"] - st["ms_mlp_fwd"] - st["ms_mlp_jac"]), flush=True)
    b.close()
