#!/usr/bin/env python3
"""The mono path's two hypotheses of ONE object, run by the UNMODIFIED reference (build container only; oracle/ref_shim.py).

src/LocalMapping_util.cc:391-406: an object that is not reconstructed yet is optimised twice -- from its pose and from the pose turned by
180 degrees about the object's up axis (`flipped_Two.col(0) *= -1; flipped_Two.col(2) *= -1`) -- and C++ keeps the result with the SMALLER
`loss`.  This records both runs (full per-iteration trace incl. `it_loss`, tools/make_golden.py's Recorder) on the decoder fitted to the
complex car family (cabin and spoiler make it fore/aft asymmetric; a rounded box would make the two losses equal), at the size SLAM
really calls the path with (250 surface points, 200 background rays), Freiburg hyper-parameters (configs/config_freiburg_001.json:15-30:
the reference's monocular cars configuration).  -> tests/golden/golden_mono_flip.npz: keys `a_*` (the detection's pose), `b_*` (flipped).

    python tools/make_golden_mono_flip.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import ref_shim  # noqa: E402
from dsp_slam_amd import synth, fixtures  # noqa: E402
import make_golden as MG  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    ref_shim.install()
    import reconstruct.optimizer as ropt
    import reconstruct.loss as rloss
    from reconstruct.utils import get_configs, get_decoder
    torch.manual_seed(0)
    tmp = tempfile.mkdtemp(prefix="dsp_flip_")
    cx_dir = fixtures.materialize_decoder_dir("complex", os.path.join(tmp, "complex_64"))
    cfg_d = MG.make_cfg(cx_dir, MG.FREIBURG, "Freiburg")
    with open(os.path.join(tmp, "cfg.json"), "w") as f:
        json.dump(cfg_d, f)
    dec = get_decoder(get_configs(os.path.join(tmp, "cfg.json")))
    for p in dec.parameters():
        p.requires_grad_(False)
    obj = synth.make_object(51, n_surface=250, n_background=200, shape="complex")
    flip = np.diag([-1.0, 1.0, -1.0, 1.0]).astype(np.float32)         # columns 0 and 2 of T_cam_obj negated (LocalMapping_util.cc:399-401)
    out = {}
    for tag, t0 in (("a_", obj["t_cam_obj_init"]), ("b_", (obj["t_cam_obj_init"] @ flip).astype(np.float32))):
        r = MG.run_recon(ropt.Optimizer, ropt, rloss, dec, cfg_d, dict(obj, t_cam_obj_init=t0), None, get_configs)
        assert bool(r["is_good"]) and "it_loss" in r
        print(tag, "loss", float(r["loss"]), "it_loss", r["it_loss"], "K", r["it_K"])
        for k, v in r.items():
            out[tag + k] = v
    la, lb = float(out["a_loss"]), float(out["b_loss"])
    print("C++ keeps hypothesis", "b (flipped)" if la > lb else "a", "; relative gap %.3f" % (abs(la - lb) / min(la, lb)))
    np.savez_compressed(os.path.join(GOLD, "golden_mono_flip.npz"), **out)


if __name__ == "__main__":
    main()
