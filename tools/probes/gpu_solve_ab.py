#!/usr/bin/env python3
"""Development aid: one KITTI-size detection (p50 of 21 runs) and one 64-object step, with a digest of the result bits -- run once per
library variant (DSPGN_LIB) to A/B a change that must not move a bit (e.g. -DSOLVE_SEPARATE_REDUCE)."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E  # noqa: E402
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "main"
eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
prm = E.gn_params()


def digest(res):
    h = hashlib.sha256()
    for a in res:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:12]


det = synth.make_object(4242, n_surface=250, n_background=200)
b = eng.batch(prm, [det["t_cam_obj_init"]], [det["pts"]], [det["rays"]], [det["depth"]])
b.run()
ts = []
for _ in range(21):
    t0 = time.perf_counter()
    b.run()
    r = b.results()
    ts.append((time.perf_counter() - t0) * 1e3)
d1 = digest(r)
b.close()
objs = synth.make_batch(64, first_seed=1, n_surface=2000, n_background=500)
bb = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
bb.run()
t0 = time.perf_counter()
bb.run()
dt = time.perf_counter() - t0
d2 = digest(bb.results())
bb.close()
pe = eng.estimate_pose_batch(prm, [np.eye(4, dtype=np.float32)], [1.0], [det["pts"]], [np.zeros(64, np.float32)])
print("%s: detection p50 %.3f ms (min %.3f) bits %s | 64-object step %.1f ms bits %s | pose-only bits %s" % (
    name, float(np.median(ts)), min(ts), d1, dt * 1e3, d2, digest([pe])))
eng.close()
