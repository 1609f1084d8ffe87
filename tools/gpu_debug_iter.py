#!/usr/bin/env python3
"""Development aid: find the GN iteration of the small golden where device and oracle H differ most and compare the
render rows at that state."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dsp_oracle as O
from dsp_slam_amd import fixtures, engine as E
dec = O.fold_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), fixtures.SPECS)
eng = E.Engine(dec.layers, dec.latent_in, dec.code_len, device=0)
g = np.load(os.path.join(ROOT, "tests/golden/golden_recon_small.npz"))
cfg = json.loads(str(g["cfg_json"]))
prm, oprm = E.params_from_configs(cfg), O.GNParams.from_configs(cfg)
b = eng.batch(prm, [g["in_t_cam_obj_init"]], [g["in_pts"]], [g["in_rays"]], [g["in_depth"]], trace=True)
b.run()
o1 = O.GNParams(num_iterations=1)
for e in range(10):
    tr = b.trace(e)
    otr = []
    O.reconstruct_object(dec, o1, None, g["in_pts"], g["in_rays"], g["in_depth"], tr["code"][0], trace=otr, t_obj_cam0=tr["t_obj_cam"][0])
    it = otr[0]
    dh = np.abs(tr["H"][0] - it["H"]).max() / np.abs(it["H"]).max()
    same = int(tr["set_sums"][0][0]) == it["vsum"] and int(tr["set_sums"][0][1]) == it["ksum"]
    print("iter %d: V %d/%d m %d/%d K %d/%d same_sets %s  relH %.2e" % (e, tr["V"][0], it["V"], tr["m"][0], it["m"], tr["K"][0], it["K"], same, dh))
    if dh > 2e-4 and same:
        t_oc, code = tr["t_obj_cam"][0], tr["code"][0]
        t_co = O._inv(t_oc); scale = O._det3_cuberoot(t_co[:3, :3])
        dmin, dmax = np.float32(t_co[2, 3] - scale), np.float32(t_co[2, 3] + scale)
        sampled = O.linspace_f32(dmin, dmax, 50)
        n_fg = g["in_depth"].shape[0]
        depth_obs = np.concatenate([g["in_depth"], np.full(g["in_rays"].shape[0] - n_fg, np.float32(1.1) * dmax, np.float32)])
        st = {}
        oj7, ojc, orr = O.compute_render_loss(dec, g["in_rays"], depth_obs, t_oc, sampled, code, 0.01, stats=st)
        (gj7, gjc, grr), gst = eng.compute_render_loss(g["in_rays"], depth_obs, t_oc, sampled, code, 0.01)
        print("   rows oracle %d gpu %d" % (oj7.shape[0], gj7.shape[0]))
        if oj7.shape == gj7.shape:
            d7 = np.abs(oj7 - gj7).max(1); rel7 = d7 / np.maximum(np.abs(oj7).max(1), 1e-12)
            worst = np.argsort(-d7)[:5]
            print("   res max diff %.3e" % np.abs(orr - grr).max())
            gx, gy = st["kept"]
            for w in worst:
                sdf_w = st["sdf"][np.where((st["valid"][0] == gx[w]) & (st["valid"][1] == gy[w]))[0][0]]
                print("   row %d (ray %d, j %d): |dJ7| %.3e rel %.2e  |J7| %.3e  de_ds %.4e  sdf %.6e  1-o_k %.3e" % (
                    w, gx[w], gy[w], d7[w], rel7[w], np.abs(oj7[w]).max(), st["de_ds"][w], sdf_w, 0.5 + sdf_w / 0.02))
            # contribution of those rows to H[0,0]
            print("   sum J0^2 oracle %.6f gpu %.6f ; top row share %.3f" % ((oj7[:, 0] ** 2).sum(), (gj7[:, 0] ** 2).sum(), (oj7[worst[0], 0] ** 2) / (oj7[:, 0] ** 2).sum()))
        break

# ---- which entries, and does the device's own depth derivation explain it?
e = 9
tr = b.trace(e)
otr = []
O.reconstruct_object(dec, o1, None, g["in_pts"], g["in_rays"], g["in_depth"], tr["code"][0], trace=otr, t_obj_cam0=tr["t_obj_cam"][0])
it = otr[0]
dH = np.abs(tr["H"][0] - it["H"])
idx = np.dstack(np.unravel_index(np.argsort(-dH.ravel())[:8], dH.shape))[0]
print("largest |dH| entries:", [(int(i), int(j), float(dH[i, j]), float(it["H"][i, j])) for i, j in idx])
print("b diff max", np.abs(tr["b"][0] - it["b"]).max(), "at", int(np.argmax(np.abs(tr["b"][0] - it["b"]))), "b scale", np.abs(it["b"]).max())
# Hs / Hr split via the stand-alone terms on the device at that state with ORACLE depths
t_oc, code = tr["t_obj_cam"][0], tr["code"][0]
j7s, jcs, rs = eng.compute_sdf_loss(g["in_pts"], t_oc, code)
oj7s, ojcs, ors = O.compute_sdf_loss(dec, g["in_pts"], t_oc, code)
print("sdf term rows: J7 diff %.3e (|J| %.3e)  Jc diff %.3e  r diff %.3e" % (np.abs(j7s - oj7s).max(), np.abs(oj7s).max(), np.abs(jcs - ojcs).max(), np.abs(rs - ors).max()))
Js = np.concatenate([oj7s, ojcs], 1); Jsd = np.concatenate([j7s, jcs], 1)
print("Hs[0,0] oracle %.6f device-rows %.6f  (x k2/M = %.4f)" % ((Js[:, 0] ** 2).sum(), (Jsd[:, 0] ** 2).sum(), 100.0 / Js.shape[0] * (Js[:, 0] ** 2).sum()))
print("H[0,0] device trace %.6f oracle %.6f" % (tr["H"][0][0, 0], it["H"][0, 0]))

# ---- device depths vs oracle depths
for e in (0, 5, 9):
    tr = b.trace(e)
    t_co = O._inv(tr["t_obj_cam"][0]); scale = O._det3_cuberoot(t_co[:3, :3])
    samp = O.linspace_f32(np.float32(t_co[2, 3] - scale), np.float32(t_co[2, 3] + scale), 50)
    dd = tr["depths"][0][:50]
    print("iter %d depths: max|dev-oracle| %.3e (spacing %.4f) dev[0] %.7f orc[0] %.7f dev[49] %.7f orc[49] %.7f" % (e, np.abs(dd - samp).max(), samp[1] - samp[0], dd[0], samp[0], dd[49], samp[49]))
