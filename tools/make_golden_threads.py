#!/usr/bin/env python3
"""Adds the reference's NO-INPUT-CHANGE spread to the recorded goldens (build container only: imports /root/reference through
oracle/ref_shim.py, modifies nothing there).

    python tools/make_golden_threads.py                      # every tests/golden/golden_recon_*.npz
    python tools/make_golden_threads.py --bench              # the traced objects of golden_bench_cfg2x64.npz

The unmodified reference is re-run on EXACTLY the recorded inputs -- same arrays, same process recipe -- with only
`torch.set_num_threads(n)` changed (n = 1 and 4; the goldens were recorded at 8, the bench golden at 6).  A thread count changes nothing
but the order in which the CPU sgemm / reductions accumulate, i.e. it is a perturbation of the LAST BIT of intermediate float32 sums
with no input change at all -- the cleanest yardstick of how far the chained 10-iteration map carries round-off (1.19e-3 pose /
6.2e-4 code on cfg2 from the thread count alone).  Stored per file: `thr_counts` (n,), `thr_t_cam_obj` (n, 4, 4), `thr_code` (n, 64) --
for the bench golden `tr<i>_thr_*` per traced object; every other array is written back unchanged.  tests/test_gpu_parity.py
(end_to_end_differences) takes the larger of this and the 1-ulp input spread (`ulps_*`, tools/make_golden_sensitivity.py).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import ref_shim  # noqa: E402
from dsp_slam_amd import fixtures, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
COUNTS = (1, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", action="store_true")
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    ref_shim.install()
    import torch
    from reconstruct.optimizer import Optimizer
    from reconstruct.utils import get_configs, get_decoder
    torch.manual_seed(0)
    tmp = tempfile.mkdtemp(prefix="dsp_thr_")
    dirs = {64: fixtures.materialize_decoder_dir("cars", os.path.join(tmp, "cars_64")),
            32: fixtures.materialize_decoder_dir("chairs32", os.path.join(tmp, "chairs_32"))}

    def optimizer_for(cfg_d):
        cfg_d = dict(cfg_d)
        if os.path.basename(cfg_d["DeepSDF_DIR"]).startswith("complex") and "complex" not in dirs:
            dirs["complex"] = fixtures.materialize_decoder_dir("complex", os.path.join(tmp, "complex_64"))
        cfg_d["DeepSDF_DIR"] = dirs["complex"] if os.path.basename(cfg_d["DeepSDF_DIR"]).startswith("complex") else dirs[cfg_d["optimizer"]["code_len"]]
        with open(os.path.join(tmp, "cfg.json"), "w") as f:
            json.dump(cfg_d, f)
        cfg = get_configs(os.path.join(tmp, "cfg.json"))
        decoder = get_decoder(cfg)
        for p in decoder.parameters():
            p.requires_grad_(False)
        return Optimizer(decoder, cfg)

    def runs(opt, t0, pts, rays, depth, code0):
        ts, cs = [], []
        for n in COUNTS:
            torch.set_num_threads(n)
            with contextlib.redirect_stdout(io.StringIO()):
                r = opt.reconstruct_object(t0.copy(), pts.copy(), rays.copy(), depth.copy(), None if code0 is None else code0.copy())
            assert r.is_good
            ts.append(np.asarray(r.t_cam_obj, np.float32))
            cs.append(np.asarray(r.code, np.float32))
        return np.stack(ts), np.stack(cs)

    if args.bench:
        path = os.path.join(GOLD, "golden_bench_cfg2x64.npz")
        g = dict(np.load(path, allow_pickle=False))
        opt = optimizer_for(json.loads(str(g["cfg_json"])))
        objs = synth.make_batch(int(g["all_t_cam_obj"].shape[0]), first_seed=int(g["first_seed"]), n_surface=int(g["n_surface"]),
                                n_background=int(g["n_background"]))
        g["thr_counts"] = np.array(COUNTS, np.int64)
        for i in [int(k) for k in g["full_objects"]]:
            o = objs[i]
            ts, cs = runs(opt, o["t_cam_obj_init"], o["pts"], o["rays"], o["depth"], None)
            g["tr%d_thr_t_cam_obj" % i], g["tr%d_thr_code" % i] = ts, cs
            print("bench object %2d: thread counts %s vs the recording: |dT| %.2e |dcode| %.2e   (1-ulp input draws: %.2e / %.2e)" % (
                i, COUNTS, np.abs(ts - g["all_t_cam_obj"][i]).max(), np.abs(cs - g["all_code"][i]).max(),
                np.abs(g["tr%d_ulps_t_cam_obj" % i] - g["all_t_cam_obj"][i]).max(), np.abs(g["tr%d_ulps_code" % i] - g["all_code"][i]).max()), flush=True)
            np.savez_compressed(path, **g)
        return
    names = args.names or sorted(f for f in os.listdir(GOLD) if f.startswith("golden_recon_") and f != "golden_recon_fail.npz")
    for name in names:
        path = os.path.join(GOLD, name)
        g = dict(np.load(path, allow_pickle=False))
        if not bool(g["is_good"]):
            continue
        opt = optimizer_for(json.loads(str(g["cfg_json"])))
        ts, cs = runs(opt, g["in_t_cam_obj_init"], g["in_pts"], g["in_rays"], g["in_depth"], g["in_code"] if "in_code" in g else None)
        g["thr_counts"], g["thr_t_cam_obj"], g["thr_code"] = np.array(COUNTS, np.int64), ts, cs
        np.savez_compressed(path, **g)
        print("%-28s thread counts %s vs the recording (8 threads): |dT| %.2e |dcode| %.2e   (1-ulp input draws: %.2e / %.2e)" % (
            name, COUNTS, np.abs(ts - g["t_cam_obj"]).max(), np.abs(cs - g["code"]).max(),
            np.abs(g["ulps_t_cam_obj"] - g["t_cam_obj"]).max(), np.abs(g["ulps_code"] - g["code"]).max()), flush=True)


if __name__ == "__main__":
    main()
