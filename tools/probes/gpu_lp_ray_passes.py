#!/usr/bin/env python3
"""Development aid (GPU box): the headline batch (64 cfg2 objects) with n front-to-back depth ranges, default path and the low-precision
compute mode: ms per run and samples decoded.  python tools/probes/gpu_lp_ray_passes.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm

eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
prm = E.gn_params()
objs = synth.make_batch(64, first_seed=1, n_surface=2000, n_background=500)
args = ([o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
b = eng.batch(prm, *args)
b.set_kernel_timing(1)
ref = {}
for compute in (0, 1):
    b.set_compute(compute)
    for n in (0, 5, 7, 10, 13, 17, 25, 50):
        b.set_ray_passes(n)
        b.run()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            b.run()
            ts.append((time.perf_counter() - t0) * 1e3)
        st = b.stats()
        res = b.results()
        if (compute, 0) not in ref:
            ref[(compute, 0)] = res
        same = all(np.array_equal(x, y) for x, y in zip(res, ref[(compute, 0)]))
        print("compute %d  ray passes %2d: %8.2f ms per run (min of 3 %8.2f)  prepass %7.2f ms / %3d launches / %.2f M samples   fwd fp32 %7.2f ms   jac %7.2f ms   same bits as automatic: %s" % (
            compute, n, float(np.median(ts)), min(ts), st["ms_mlp_prepass"], st["n_mlp_prepass_launches"], st["n_prepass_points"] / 1e6, st["ms_mlp_fwd"], st["ms_mlp_jac"], same), flush=True)
b.close()
eng.close()
