#!/usr/bin/env python3
"""Development aid: p50 / min of ONE real-KITTI-size detection (resident batch, run + results), for A/B runs of library variants
(DSPGN_LIB=...): python tools/gpu_detection_p50.py [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
o = synth.make_object(4242, n_surface=250, n_background=200)
b = eng.batch(E.gn_params(), [o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
for _ in range(3):
    b.run(); r = b.results()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); b.run(); r = b.results(); ts.append((time.perf_counter() - t0) * 1e3)
st = b.stats()
import hashlib
print("%s: p50 %.3f ms  min %.3f  cluster tiles %d  fallback %d  digest %s" % (os.path.basename(os.environ.get("DSPGN_LIB", "libdspgn.so")), np.median(ts), np.min(ts),
      st["n_cluster_tiles"], st["cluster_fallback"], hashlib.sha1(b"".join(np.ascontiguousarray(x).tobytes() for x in r)).hexdigest()[:12]))
