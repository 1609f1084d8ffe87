#!/usr/bin/env python3
"""Adds the reference's own round-off spread to the recorded GN goldens (build container only: imports /root/reference
through oracle/ref_shim.py).

For every tests/golden/golden_recon_*.npz that holds a good result, the UNMODIFIED reference is re-run on N_DRAWS copies of the
recorded inputs in which every element of pts / rays / depth is moved to an adjacent float32 (up or down, seeded): a
perturbation of the size any re-ordering of one float32 sum produces.  The final poses / codes of those runs are stored as
`ulps_t_cam_obj` (N, 4, 4) and `ulps_code` (N, 64); every other array of the file is written back unchanged.  The end-to-end
parity test bounds |device - reference| by the largest |perturbed reference - reference| (tests/test_gpu_parity.py).
The unperturbed re-run must reproduce the recorded result bit for bit (checked)."""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from dsp_slam_amd import fixtures  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
N_DRAWS = int(os.environ.get("DSP_SENS_DRAWS", "8"))      # the chaotic full-size fixtures (cfg2, complex) carry 32: eight draws do not bound a heavy-tailed spread


def jiggle(a, rng):
    a = np.ascontiguousarray(a, np.float32)
    up = rng.integers(0, 2, size=a.shape).astype(bool)
    return np.where(up, np.nextafter(a, np.float32(np.inf)), np.nextafter(a, np.float32(-np.inf))).astype(np.float32)


def main():
    ref_shim.install()
    import torch
    from reconstruct.optimizer import Optimizer
    from reconstruct.utils import get_configs, get_decoder
    torch.manual_seed(0)
    tmp = tempfile.mkdtemp(prefix="dsp_sens_")
    dirs = {64: fixtures.materialize_decoder_dir("cars", os.path.join(tmp, "cars_64")),
            32: fixtures.materialize_decoder_dir("chairs32", os.path.join(tmp, "chairs_32"))}
    names = sys.argv[1:] or sorted(f for f in os.listdir(GOLD) if f.startswith("golden_recon_") and f != "golden_recon_fail.npz")
    for name in names:
        path = os.path.join(GOLD, name)
        g = dict(np.load(path, allow_pickle=False))
        if not bool(g["is_good"]):
            continue
        cfg_d = json.loads(str(g["cfg_json"]))
        # 64-D goldens use the cars fixture, 32-D ones chairs32 -- unless the recorded directory names another fixture (complex_64)
        if os.path.basename(cfg_d["DeepSDF_DIR"]).startswith("complex") and "complex" not in dirs:
            dirs["complex"] = fixtures.materialize_decoder_dir("complex", os.path.join(tmp, "complex_64"))
        cfg_d["DeepSDF_DIR"] = dirs["complex"] if os.path.basename(cfg_d["DeepSDF_DIR"]).startswith("complex") else dirs[cfg_d["optimizer"]["code_len"]]
        with open(os.path.join(tmp, "cfg.json"), "w") as f:
            json.dump(cfg_d, f)
        cfg = get_configs(os.path.join(tmp, "cfg.json"))
        decoder = get_decoder(cfg)
        for p in decoder.parameters():
            p.requires_grad_(False)
        opt = Optimizer(decoder, cfg)
        code0 = g["in_code"] if "in_code" in g else None

        def run(pts, rays, depth):
            with contextlib.redirect_stdout(io.StringIO()):
                return opt.reconstruct_object(g["in_t_cam_obj_init"].copy(), pts.copy(), rays.copy(), depth.copy(),
                                              None if code0 is None else code0.copy())

        base = run(g["in_pts"], g["in_rays"], g["in_depth"])
        assert np.array_equal(np.asarray(base.t_cam_obj, np.float32), g["t_cam_obj"]), name + ": the recorded result does not reproduce"
        rng = np.random.default_rng(20260925)
        ts, cs = [], []
        for _ in range(N_DRAWS):
            r = run(jiggle(g["in_pts"], rng), jiggle(g["in_rays"], rng), jiggle(g["in_depth"], rng))
            assert r.is_good
            ts.append(np.asarray(r.t_cam_obj, np.float32))
            cs.append(np.asarray(r.code, np.float32))
        g["ulps_t_cam_obj"] = np.stack(ts)
        g["ulps_code"] = np.stack(cs)
        np.savez_compressed(path, **g)
        print(name, "max |dT| over draws %.3e (single recorded draw %.3e)   max |dcode| %.3e (%.3e)" % (
            np.abs(g["ulps_t_cam_obj"] - g["t_cam_obj"]).max(), np.abs(g["ulp_t_cam_obj"] - g["t_cam_obj"]).max(),
            np.abs(g["ulps_code"] - g["code"]).max(), np.abs(g["ulp_code"] - g["code"]).max()), flush=True)


if __name__ == "__main__":
    main()
