#!/usr/bin/env python3
"""Quick on-GPU sanity + timing probe (development aid): decoder parity on a few sizes, one cfg2 batch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dsp_oracle as O
from dsp_slam_amd import fixtures, synth, engine as E

dec = O.fold_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), fixtures.SPECS)
eng = E.Engine(dec.layers, dec.latent_in, dec.code_len, device=0)
rng = np.random.default_rng(0)
code = (rng.normal(size=64) * 0.2).astype(np.float32)
for n in (16, 64, 1000):
    pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    out = eng.decode_sdf(code, pts)
    ref = O.decode_sdf(dec, code, pts)
    print("fwd n=%d max|diff| %.3e  (|ref| max %.3f)" % (n, np.abs(out - ref).max(), np.abs(ref).max()), flush=True)
    sdf, grad = eng.sdf_jacobian(code, pts)
    y, g = O.get_batch_sdf_jacobian(dec, code, pts)
    print("jac n=%d sdf diff %.3e grad diff %.3e (|g| max %.3f) code-part %.3e xyz-part %.3e" % (
        n, np.abs(sdf - y).max(), np.abs(grad - g).max(), np.abs(g).max(), np.abs(grad[:, :64] - g[:, :64]).max(),
        np.abs(grad[:, 64:] - g[:, 64:]).max()), flush=True)
n = 64 * 256 * 8
pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
for rep in range(3):
    t0 = time.time(); eng.decode_sdf(code, pts); dt = time.time() - t0
    print("decode %d pts: %.1f ms wall incl. copies -> %.1f TFLOP/s (fwd 3.671 MFLOP/pt)" % (n, dt * 1e3, n * 3.67104e6 / dt / 1e12), flush=True)

nobj = int(os.environ.get("PROBE_OBJS", "8"))
objs = synth.make_batch(nobj, first_seed=1, n_surface=2000, n_background=500)
prm = E.gn_params()
b = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs])
for rep in range(3):
    t0 = time.time(); b.run(); dt = time.time() - t0
    st = b.stats()
    jflops = st["n_jac_points"] * 7.34208e6 + st["n_render_rows"] * 3.67104e6
    flops = st["n_fwd_points"] * 3.67104e6 + jflops
    print("cfg2 x%d: %.1f ms wall, %.2f obj/s; events total %.1f ms, mlp fwd %.1f ms (%d launches) jac %.1f ms (%d); "
          "V+K pts %.3g/%.3g; alg %.2f TFLOP -> %.1f TFLOP/s overall, fwd kernel %.1f TFLOP/s, jac kernel %.1f TFLOP/s" % (
              nobj, dt * 1e3, nobj / dt, st["ms_total"], st["ms_mlp_fwd"], st["n_mlp_fwd_launches"], st["ms_mlp_jac"], st["n_mlp_jac_launches"],
              st["n_fwd_points"], st["n_jac_points"], flops / 1e12, flops / dt / 1e12,
              st["n_fwd_points"] * 3.67104e6 / (st["ms_mlp_fwd"] * 1e-3) / 1e12, jflops / (st["ms_mlp_jac"] * 1e-3) / 1e12), flush=True)
t, c, l, s = b.results()
print("status", s, "loss", l)
for i, o in enumerate(objs[:4]):
    print(" obj %d t-err %.4f -> %.4f" % (i, np.linalg.norm(o["t_cam_obj_init"][:3, 3] - o["t_cam_obj_gt"][:3, 3]), np.linalg.norm(t[i][:3, 3] - o["t_cam_obj_gt"][:3, 3])))
