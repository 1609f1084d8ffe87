#!/usr/bin/env python3
"""Development aid: uniform vs non-uniform depth ranges for the front-to-back passes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9)
eng = E.Engine(layers, [4], 64, device=0)
prm = E.gn_params()
def uni(n): return [round(50 * p / n) for p in range(n + 1)]
def quad(n, a): return [int(round(50 * (a * p / n + (1 - a) * (p / n) ** 2))) for p in range(n)] + [50]
for B in (32, 1):
    objs = synth.make_batch(B, first_seed=1, n_surface=2000, n_background=500)
    b = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
    cands = {"auto": None}
    for n in ((8, 10, 12) if B > 1 else (2, 3)):
        cands["uniform%d" % n] = uni(n)
        for a in (0.7, 0.55, 0.4):
            cands["quad%d_%.2f" % (n, a)] = quad(n, a)
    cands["hand10"] = [0, 4, 7, 10, 13, 16, 19, 22, 26, 34, 50]
    if B == 1:
        cands["hand2"] = [0, 18, 50]; cands["hand3"] = [0, 12, 22, 50]; cands["hand2b"] = [0, 22, 50]
    for name, bd in cands.items():
        if bd is None: b.set_ray_passes(0)
        else: b.set_ray_pass_bounds(bd)
        b.run()
        ts = []
        for _ in range(3 if B > 1 else 5):
            t0 = time.perf_counter(); b.run(); ts.append(time.perf_counter() - t0)
        st = b.stats()
        print("B=%d %-14s %s: %.1f ms -> %.2f obj/s ; evaluated %.1f%%" % (B, name, bd, np.median(ts) * 1e3, B / np.median(ts), 100 * st["n_fwd_points"] / st["n_insphere_points"]), flush=True)
    b.close()
