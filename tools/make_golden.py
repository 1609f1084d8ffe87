#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference Python hot path on CPU.

Runs only in the build container (imports /root/reference through oracle/ref_shim.py).
The reference has no tests or golden vectors; these files are what pins
oracle/dsp_oracle.py, and through it the HIP path, to the reference's behaviour.

    python tools/make_golden.py            # all files (cfg2 takes ~15 s of reference time)

Per-iteration internals are captured without touching reference code, by wrapping the names the
reference module looks up at call time (reconstruct.optimizer.compute_* / exp_sim3,
reconstruct.loss.decode_sdf) and torch.inverse / torch.mv while a run is in flight.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from dsp_slam_amd import synth, fixtures  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

KITTI = dict(k1=1.0, k2=100.0, k3=0.25, k4=1e7, b1=0.20, b2=0.025, num_iterations=10,
             learning_rate=1.0, scale_damping=1.0)
REDWOOD = dict(k1=10.0, k2=100.0, k3=2.5, k4=0.0, b1=0.20, b2=0.02, num_iterations=5,
               learning_rate=1.0, scale_damping=100.0)
FREIBURG = dict(k1=1.0, k2=100.0, k3=0.5, k4=0.0, b1=0.20, b2=0.025, num_iterations=5,      # configs/config_freiburg_001.json:15-30
                learning_rate=1.0, scale_damping=100.0)


def make_cfg(deepsdf_dir, joint, data_type="KITTI", code_len=64):
    return {
        "data_type": data_type, "DeepSDF_DIR": deepsdf_dir, "voxels_dim": 32,
        "optimizer": {"code_len": code_len, "num_depth_samples": 50, "cut_off_threshold": 0.01,
                      "joint_optim": dict(joint),
                      "pose_only_optim": {"num_iterations": 5, "learning_rate": 1.0}},
    }


class Recorder(object):
    """Wraps reference-module names + torch.inverse/mv to record one GN trace."""

    def __init__(self, ropt, rloss, th, k1=None, k2=None):
        self.ropt, self.rloss, self.th = ropt, rloss, th
        self.k1, self.k2 = k1, k2          # with both given, `loss = k1 * render_loss + k2 * sdf_loss` (optimizer.py:155) is recorded per iteration
        self.iters = []
        self.cur = None

    def __enter__(self):
        ro, rl = self.ropt, self.rloss
        self.saved = dict(sdf=ro.compute_sdf_loss, rend=ro.compute_render_loss, rot=ro.compute_rotation_loss_sim3,
                          exp=ro.exp_sim3, dec=rl.decode_sdf, inv=torch.inverse, mv=torch.mv, lin=torch.linspace, rob=ro.get_robust_res)
        rec = self
        rec.pending_depths = None

        def w_lin(*a, **k):          # the depth samples of the iteration about to start (optimizer.py:125)
            out = rec.saved["lin"](*a, **k)
            rec.pending_depths = out.clone().numpy()
            return out

        def w_sdf(decoder, pts, t_obj_cam, z):
            rec.cur = dict(t_obj_cam=t_obj_cam.clone().numpy(), code=z.clone().cpu().numpy())
            if rec.pending_depths is not None:
                rec.cur["depths"] = rec.pending_depths
            rec.iters.append(rec.cur)
            out = rec.saved["sdf"](decoder, pts, t_obj_cam, z)
            rec.cur["res_sdf_absmax"] = float(out[2].abs().max())
            return out

        def w_dec(decoder, z, x, *a, **k):
            out = rec.saved["dec"](decoder, z, x, *a, **k)
            if rec.cur is not None and "V" not in rec.cur:
                rec.cur["V"] = int(x.shape[0])
                s = out.reshape(-1)
                rec.cur["m"] = int(((s > -rec.th) & (s < rec.th)).sum())
            return out

        def w_rend(*a, **k):
            out = rec.saved["rend"](*a, **k)
            rec.cur["K"] = -1 if out is None else int(out[0].shape[0])
            return out

        def w_rob(res, b_):           # optimizer.py:133 (surface term, first call of an iteration) and :147 (render term, second call)
            out = rec.saved["rob"](res, b_)
            if rec.cur is not None:
                if "loss_sdf" not in rec.cur:
                    rec.cur["loss_sdf"] = out[1].clone()
                elif "loss_render" not in rec.cur:
                    rec.cur["loss_render"] = out[1].clone()
                    if rec.k1 is not None:      # the reference's own expression on the reference's own operands (optimizer.py:155)
                        rec.cur["loss"] = np.float32(float(rec.k1 * rec.cur["loss_render"] + rec.k2 * rec.cur["loss_sdf"]))
            return out

        def w_inv(x):
            out = rec.saved["inv"](x)
            if rec.cur is not None and x.shape[0] > 8:
                rec.cur["H"] = x.clone().numpy()
            return out

        def w_mv(a, b):
            out = rec.saved["mv"](a, b)
            if rec.cur is not None and a.shape[0] > 8:
                rec.cur["b"] = b.clone().numpy()
                rec.cur["dx"] = out.clone().numpy()
            return out

        ro.compute_sdf_loss, ro.compute_render_loss, rl.decode_sdf = w_sdf, w_rend, w_dec
        ro.get_robust_res = w_rob
        torch.inverse, torch.mv, torch.linspace = w_inv, w_mv, w_lin
        return self

    def __exit__(self, *exc):
        ro, rl = self.ropt, self.rloss
        ro.compute_sdf_loss, ro.compute_render_loss = self.saved["sdf"], self.saved["rend"]
        rl.decode_sdf = self.saved["dec"]
        ro.get_robust_res = self.saved["rob"]
        torch.inverse, torch.mv, torch.linspace = self.saved["inv"], self.saved["mv"], self.saved["lin"]

    def pack(self, prefix=""):
        out = {}
        keys = ["t_obj_cam", "code", "H", "b", "dx", "depths"]
        full = [it for it in self.iters if "dx" in it]
        for k in keys:
            if full:
                out[prefix + "it_" + k] = np.stack([it[k] for it in full]).astype(np.float32)
        if full and all("loss" in it for it in full):
            out[prefix + "it_loss"] = np.array([it["loss"] for it in full], np.float32)
            out[prefix + "it_loss_sdf"] = np.array([float(it["loss_sdf"]) for it in full], np.float32)
            out[prefix + "it_loss_render"] = np.array([float(it["loss_render"]) for it in full], np.float32)
        out[prefix + "it_V"] = np.array([it.get("V", -1) for it in self.iters], np.int64)
        out[prefix + "it_m"] = np.array([it.get("m", -1) for it in self.iters], np.int64)
        out[prefix + "it_K"] = np.array([it.get("K", -1) for it in self.iters], np.int64)
        return out


def run_recon(Optimizer, ropt, rloss, decoder, cfg_dict, obj, code=None, get_configs=None):
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(cfg_dict, f)
    cfg = get_configs(f.name)
    os.unlink(f.name)
    opt = Optimizer(decoder, cfg)
    j = cfg_dict["optimizer"]["joint_optim"]
    with Recorder(ropt, rloss, cfg_dict["optimizer"]["cut_off_threshold"], j["k1"], j["k2"]) as rec:
        rst = opt.reconstruct_object(obj["t_cam_obj_init"].copy(), obj["pts"].copy(), obj["rays"].copy(),
                                     obj["depth"].copy(), None if code is None else code.copy())
    out = rec.pack()
    # reference self-sensitivity: the same run with every surface point moved by one float32 ulp.  The 10-iteration
    # map is discontinuous in its ragged sets (V, m, K), so this is the yardstick for end-to-end comparisons.
    if rst.is_good:
        import contextlib, io
        pts_ulp = (obj["pts"].astype(np.float64) * (1 + 1.2e-7)).astype(np.float32)
        with contextlib.redirect_stdout(io.StringIO()):
            rst2 = opt.reconstruct_object(obj["t_cam_obj_init"].copy(), pts_ulp, obj["rays"].copy(), obj["depth"].copy(),
                                          None if code is None else code.copy())
        if rst2.is_good:
            out["ulp_t_cam_obj"] = np.asarray(rst2.t_cam_obj, np.float32)
            out["ulp_code"] = np.asarray(rst2.code, np.float32)
    out["is_good"] = np.array(bool(rst.is_good))
    out["loss"] = np.array(float(rst.loss), np.float32)
    if rst.is_good and "it_loss" in out:      # the returned loss IS the last iteration's (optimizer.py:155,200-203)
        assert out["it_loss"][-1] == out["loss"], (out["it_loss"][-1], out["loss"])
    if rst.is_good:
        out["t_cam_obj"] = np.asarray(rst.t_cam_obj, np.float32)
        out["code"] = np.asarray(rst.code, np.float32)
    for k in ("t_cam_obj_init", "pts", "rays", "depth", "t_cam_obj_gt", "code_gt"):
        out["in_" + k] = obj[k]
    if code is not None:
        out["in_code"] = code
    out["cfg_json"] = np.array(json.dumps(cfg_dict))
    return out


def main():
    only = set(sys.argv[1:])
    ref_shim.install()
    import reconstruct.optimizer as ropt
    import reconstruct.loss as rloss
    import reconstruct.loss_utils as rlu
    from reconstruct.utils import get_configs, get_decoder

    torch.manual_seed(0)
    tmp = tempfile.mkdtemp(prefix="dsp_fixture_")
    cars_dir = fixtures.materialize_decoder_dir("cars", os.path.join(tmp, "cars_64"))
    cfg_kitti = make_cfg(cars_dir, KITTI)
    with open(os.path.join(tmp, "cfg.json"), "w") as f:
        json.dump(cfg_kitti, f)
    cfg = get_configs(os.path.join(tmp, "cfg.json"))
    decoder = get_decoder(cfg)               # the reference loader on the reference on-disk format
    for p in decoder.parameters():
        p.requires_grad_(False)              # results unaffected; skips the wasted weight-grad work
    rng = np.random.default_rng(7)

    def want(name):
        return not only or name in only

    # ---- 0. MeshExtractor voxel grid (utils.py:97-116): under torch >= 1.6 `LongTensor / int` is TRUE division, so the
    #         reference's grid is sheared (y index gains z / N, x index gains y / N + z / N^2) -- recorded as it is
    if want("grid"):
        from reconstruct.utils import create_voxel_grid
        out = {}
        for n in (4, 16):
            out["grid_%d" % n] = create_voxel_grid(n).numpy()
        for n in (32, 64, 128):
            gr = create_voxel_grid(n).numpy()
            out["sample_%d" % n] = gr[::997].copy()
            out["sum_%d" % n] = gr.astype(np.float64).sum(0)
        np.savez_compressed(os.path.join(GOLD, "golden_voxel_grid.npz"), **out)

    # ---- A. decoder forward / input-Jacobian -------------------------------------------------
    if want("decoder"):
        n = 96
        code = (rng.normal(size=64) * 0.1).astype(np.float32)
        code[:3] = [0.2, -0.3, 0.1]
        pts = rng.uniform(-0.9, 0.9, size=(n, 3)).astype(np.float32)
        x = np.concatenate([np.broadcast_to(code, (n, 64)), pts], -1).astype(np.float32)
        with torch.no_grad():
            y = decoder(torch.from_numpy(x)).squeeze(-1).numpy()
        yj, gj = rlu.get_batch_sdf_jacobian(decoder, torch.from_numpy(code), torch.from_numpy(pts), 1)
        sdf = rlu.decode_sdf(decoder, torch.from_numpy(code), torch.from_numpy(pts)).numpy()
        np.savez_compressed(os.path.join(GOLD, "golden_decoder.npz"), code=code, pts=pts, y=y,
                            y_jac=yj.reshape(-1).numpy(), grad=gj.reshape(n, 67).numpy(), sdf=sdf)
        print("decoder: |y| max", np.abs(y).max(), " grad absmax", np.abs(gj.numpy()).max())

    # ---- B. residual terms, Lie helpers, Huber -----------------------------------------------
    if want("terms"):
        obj = synth.make_object(3, n_surface=64, n_background=32)
        t_obj_cam = torch.inverse(torch.from_numpy(obj["t_cam_obj_init"]))
        z = torch.from_numpy((rng.normal(size=64) * 0.05).astype(np.float32))
        j7, jc, r = rloss.compute_sdf_loss(decoder, torch.from_numpy(obj["pts"]), t_obj_cam, z)
        t_co = torch.inverse(t_obj_cam)
        scale = torch.det(t_co[:3, :3]) ** (1 / 3)
        dmin, dmax = t_co[2, 3] - 1.0 * scale, t_co[2, 3] + 1.0 * scale
        sampled = torch.linspace(dmin, dmax, 50)
        depth_obs = torch.from_numpy(np.concatenate([obj["depth"], np.zeros(32, np.float32)]))
        depth_obs[64:] = 1.1 * dmax
        rj7, rjc, rr = rloss.compute_render_loss(decoder, torch.from_numpy(obj["rays"]), depth_obs, t_obj_cam,
                                                 sampled, z, th=0.01)
        out = dict(t_obj_cam=t_obj_cam.numpy(), code=z.numpy(), pts=obj["pts"], rays=obj["rays"],
                   depth_obs=depth_obs.numpy(), sampled=sampled.numpy(),
                   sdf_j7=j7.reshape(-1, 7).numpy(), sdf_jc=jc.reshape(-1, 64).numpy(), sdf_r=r.reshape(-1).numpy(),
                   ren_j7=rj7.reshape(-1, 7).numpy(), ren_jc=rjc.reshape(-1, 64).numpy(), ren_r=rr.reshape(-1).numpy())
        # rotation prior: upright (zero branch), slightly tilted, arbitrary
        rots = []
        for i, (ax, ang) in enumerate([((1, 0, 0), 0.0), ((1, 0, 0), 0.05), ((0.3, 0.2, 0.9), 0.7)]):
            d = np.zeros(7, np.float32)
            d[3:6] = np.array(ax, np.float32) / np.linalg.norm(ax) * ang
            t = torch.mm(rlu.exp_sim3(torch.from_numpy(d)), t_obj_cam)
            jr, rres = rloss.compute_rotation_loss_sim3(t.clone())
            rots.append((t.numpy(), jr.numpy(), np.float32(rres)))
        out["rot_t"] = np.stack([a for a, _, _ in rots])
        out["rot_j"] = np.stack([b for _, b, _ in rots])
        out["rot_r"] = np.array([c for _, _, c in rots], np.float32)
        # exp maps incl. the s <= 1e-8 quirk and the theta ~ 0 branches
        xs = [np.array(v, np.float32) for v in [
            [0.1, -0.2, 0.3, 0.02, -0.01, 0.03, 0.05], [0.1, -0.2, 0.3, 0.02, -0.01, 0.03, -0.05],
            [0.1, -0.2, 0.3, 0.02, -0.01, 0.03, 0.0], [0.1, -0.2, 0.3, 0.0, 0.0, 0.0, 0.04],
            [0.1, -0.2, 0.3, 0.0, 0.0, 0.0, 0.0], [0.5, 0.1, -0.7, 1.2, -0.4, 0.8, 0.3],
            [-0.01, 0.004, 0.02, 1e-4, -2e-4, 5e-5, 1e-3]]]
        out["exp_x"] = np.stack(xs)
        out["exp_sim3"] = np.stack([rlu.exp_sim3(torch.from_numpy(v)).numpy() for v in xs])
        out["exp_se3"] = np.stack([rlu.exp_se3(torch.from_numpy(v[:6])).numpy() for v in xs])
        res = torch.from_numpy(np.array([0.0, 0.01, -0.02, 0.025, -0.3, 0.2, 1e-6, -0.0251], np.float32))
        for b_ in (0.025, 0.2):
            rrb, lossb, wb = rlu.get_robust_res(res.clone(), b_)
            out["huber_%g_rr" % b_] = rrb.reshape(-1).numpy()
            out["huber_%g_loss" % b_] = np.float32(lossb)
            out["huber_%g_w" % b_] = wb.reshape(-1).numpy()
        out["huber_res"] = res.numpy()
        np.savez_compressed(os.path.join(GOLD, "golden_terms.npz"), **out)
        print("terms: sdf rows", j7.shape[0], "render rows K =", rj7.shape[0])

    # ---- C. full GN traces -------------------------------------------------------------------
    def recon(name, obj, cfg_dict, code=None, dec=decoder):
        out = run_recon(ropt.Optimizer, ropt, rloss, dec, cfg_dict, obj, code, get_configs)
        np.savez_compressed(os.path.join(GOLD, name), **out)
        print(name, "is_good", out["is_good"], "loss", out["loss"], "V", out["it_V"], "K", out["it_K"])
        if out["is_good"]:
            gt = obj["t_cam_obj_gt"]
            print("   t-err init %.4f -> final %.4f ; |code-gt|max %.3f" % (
                np.linalg.norm(obj["t_cam_obj_init"][:3, 3] - gt[:3, 3]),
                np.linalg.norm(out["t_cam_obj"][:3, 3] - gt[:3, 3]),
                np.abs(out["code"] - obj["code_gt"]).max()))

    if want("small"):
        recon("golden_recon_small.npz", synth.make_object(11, n_surface=200, n_background=50), cfg_kitti)
    if want("redwood"):
        obj = synth.make_object(12, n_surface=160, n_background=40)
        code0 = np.zeros(64, np.float32)
        code0[:3] = obj["code_gt"][:3] * 0.5
        recon("golden_recon_redwood.npz", obj, make_cfg(cars_dir, REDWOOD, "Redwood"), code=code0)
    if want("freiburg"):
        recon("golden_recon_freiburg.npz", synth.make_object(15, n_surface=180, n_background=60), make_cfg(cars_dir, FREIBURG, "Freiburg"))
    if want("cfg1"):
        c1 = make_cfg(cars_dir, dict(KITTI, num_iterations=5))
        recon("golden_recon_cfg1.npz", synth.make_object(0, n_surface=500, n_background=0), c1)
    if want("cfg2"):
        recon("golden_recon_cfg2.npz", synth.make_object(1, n_surface=2000, n_background=500), cfg_kitti)
    if want("fail"):
        # failure path: random-weight decoder => no zero crossing => K = 0 => NaN => is_good False
        rdir = os.path.join(tmp, "rand_64")
        os.makedirs(os.path.join(rdir, "ModelParameters"))
        with open(os.path.join(rdir, "specs.json"), "w") as f:
            json.dump(fixtures.SPECS, f)
        sd = fixtures.random_state_dict(5)
        torch.save({"epoch": 0, "model_state_dict": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}},
                   os.path.join(rdir, "ModelParameters", "latest.pth"))
        cfg_r = make_cfg(rdir, KITTI)
        with open(os.path.join(tmp, "cfg_r.json"), "w") as f:
            json.dump(cfg_r, f)
        dec_r = get_decoder(get_configs(os.path.join(tmp, "cfg_r.json")))
        recon("golden_recon_fail.npz", synth.make_object(13, n_surface=100, n_background=30), cfg_r, dec=dec_r)

    # ---- C2. a second decoder: 32-D codes (the Redwood chairs option, LocalMapping_util.cc:415-423), fitted to a different shape family
    if want("chairs32"):
        ch_dir = fixtures.materialize_decoder_dir("chairs32", os.path.join(tmp, "chairs_32"))
        cfg_ch = make_cfg(ch_dir, REDWOOD, "Redwood", code_len=32)
        with open(os.path.join(tmp, "cfg_ch.json"), "w") as f:
            json.dump(cfg_ch, f)
        dec_ch = get_decoder(get_configs(os.path.join(tmp, "cfg_ch.json")))
        for p_ in dec_ch.parameters():
            p_.requires_grad_(False)
        n = 96
        code = (rng.normal(size=32) * 0.1).astype(np.float32)
        code[:3] = [0.2, -0.3, 0.1]
        pts = rng.uniform(-0.9, 0.9, size=(n, 3)).astype(np.float32)
        yj, gj = rlu.get_batch_sdf_jacobian(dec_ch, torch.from_numpy(code), torch.from_numpy(pts), 1)
        sdf = rlu.decode_sdf(dec_ch, torch.from_numpy(code), torch.from_numpy(pts)).numpy()
        np.savez_compressed(os.path.join(GOLD, "golden_decoder_chairs32.npz"), code=code, pts=pts, y_jac=yj.reshape(-1).numpy(),
                            grad=gj.reshape(n, 35).numpy(), sdf=sdf)
        obj = synth.make_object(21, n_surface=220, n_background=60, code_len=32, half=synth.CHAIR_HALF)
        recon("golden_recon_chairs32.npz", obj, cfg_ch, dec=dec_ch)
    # ---- C3. BASELINE configs[4] at full size: 4000 surface points + 500 background rays, the 32-D chairs decoder, Redwood hyper-parameters
    if want("cfg5"):
        ch_dir = fixtures.materialize_decoder_dir("chairs32", os.path.join(tmp, "chairs_32b"))
        cfg_ch = make_cfg(ch_dir, REDWOOD, "Redwood", code_len=32)
        with open(os.path.join(tmp, "cfg_ch5.json"), "w") as f:
            json.dump(cfg_ch, f)
        dec_ch = get_decoder(get_configs(os.path.join(tmp, "cfg_ch5.json")))
        for p_ in dec_ch.parameters():
            p_.requires_grad_(False)
        obj = synth.make_object(31, n_surface=4000, n_background=500, code_len=32, half=synth.CHAIR_HALF)
        recon("golden_recon_cfg5.npz", obj, cfg_ch, dec=dec_ch)

    # ---- C4. a decoder of realistic geometric complexity: fitted to synth.complex_car_sdf -- body + cabin + four wheels + a
    #          thin spoiler plate, shape parameters driven by ALL 64 code dimensions (tools/fit_decoder_gpu.py --shape complex) -- one
    #          cfg2-size object under the KITTI hyper-parameters, full per-iteration trace
    if want("complex") and os.path.exists(fixtures.fixture_path("complex")):
        cx_dir = fixtures.materialize_decoder_dir("complex", os.path.join(tmp, "complex_64"))
        cfg_cx = make_cfg(cx_dir, KITTI)
        with open(os.path.join(tmp, "cfg_cx.json"), "w") as f:
            json.dump(cfg_cx, f)
        dec_cx = get_decoder(get_configs(os.path.join(tmp, "cfg_cx.json")))
        for p_ in dec_cx.parameters():
            p_.requires_grad_(False)
        n = 96
        obj = synth.make_object(41, n_surface=2000, n_background=500, shape="complex")
        code = (obj["code_gt"] * 0.8).astype(np.float32)
        pts = rng.uniform(-0.9, 0.9, size=(n, 3)).astype(np.float32)
        yj, gj = rlu.get_batch_sdf_jacobian(dec_cx, torch.from_numpy(code), torch.from_numpy(pts), 1)
        sdf = rlu.decode_sdf(dec_cx, torch.from_numpy(code), torch.from_numpy(pts)).numpy()
        truth = synth.complex_sdf(pts, code)
        print("complex decoder vs the analytic field at 96 points: max |d| %.3e (clamped to +-0.1: %.3e)" % (
            np.abs(sdf - truth).max(), np.abs(np.clip(sdf, -0.1, 0.1) - np.clip(truth, -0.1, 0.1)).max()))
        np.savez_compressed(os.path.join(GOLD, "golden_decoder_complex.npz"), code=code, pts=pts, y_jac=yj.reshape(-1).numpy(),
                            grad=gj.reshape(n, 67).numpy(), sdf=sdf)
        recon("golden_recon_complex.npz", obj, cfg_cx, dec=dec_cx)

    # ---- D. pose-only optimiser --------------------------------------------------------------
    if want("pose"):
        obj = synth.make_object(14, n_surface=300, n_background=0)
        s = float(obj["scale"])
        t_se3 = obj["t_cam_obj_init"].copy()
        t_se3[:3, :3] /= s
        code = np.zeros(64, np.float32)
        code[:3] = obj["code_gt"][:3]
        opt = ropt.Optimizer(decoder, cfg)
        rst = opt.estimate_pose_cam_obj(t_se3.copy(), s, obj["pts"].copy(), code.copy())
        np.savez_compressed(os.path.join(GOLD, "golden_pose_only.npz"), t_co_se3=t_se3, scale=np.float32(s),
                            pts=obj["pts"], code=code, out=rst.numpy(), t_cam_obj_gt=obj["t_cam_obj_gt"])
        gt = obj["t_cam_obj_gt"]
        print("pose-only: t-err %.4f -> %.4f" % (np.linalg.norm(t_se3[:3, 3] - gt[:3, 3]),
                                                 np.linalg.norm(rst.numpy()[:3, 3] - gt[:3, 3])))


if __name__ == "__main__":
    main()
