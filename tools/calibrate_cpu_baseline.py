#!/usr/bin/env python3
"""Calibrates bench.py's travelling CPU baseline against the unmodified reference (build container only).

bench.py times oracle/torch_baseline.py on the GPU box (no checkout of the reference there).  This script times the SAME cfg2
object through (a) the unmodified reference, reconstruct/optimizer.py via oracle/ref_shim.py, decoder parameters left requiring
grad as deep_sdf/workspace.py:213-221 leaves them, and (b) oracle/torch_baseline.py, alternating, and writes the ratio to
profiles/r04_cpu_baseline_calibration.md + profiles/cpu_baseline_calibration.json (read by bench.py for `calibrated_vs_reference`).
Also checks that the two agree on the result (same algorithm: differences are round-off of identical torch ops => bit-identical
unless thread scheduling differs).

    python tools/calibrate_cpu_baseline.py [--reps 3]
"""
import argparse
import contextlib
import io
import json
import os
import statistics
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim, torch_baseline as TB, dsp_oracle as O  # noqa: E402
from dsp_slam_amd import fixtures, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(ncpu)
    ref_shim.install(force_cpu=True)
    from reconstruct.optimizer import Optimizer
    from reconstruct.utils import get_configs, get_decoder

    tmp = tempfile.mkdtemp(prefix="dsp_cal_")
    ddir = fixtures.materialize_decoder_dir("cars", os.path.join(tmp, "cars_64"))
    cfg_d = {"data_type": "KITTI", "DeepSDF_DIR": ddir, "voxels_dim": 32,
             "optimizer": {"code_len": 64, "num_depth_samples": 50, "cut_off_threshold": 0.01,
                           "joint_optim": dict(k1=1.0, k2=100.0, k3=0.25, k4=1e7, b1=0.2, b2=0.025, num_iterations=10, learning_rate=1.0,
                                               scale_damping=1.0),
                           "pose_only_optim": {"num_iterations": 5, "learning_rate": 1.0}}}
    with open(os.path.join(tmp, "cfg.json"), "w") as f:
        json.dump(cfg_d, f)
    cfg = get_configs(os.path.join(tmp, "cfg.json"))
    ref_dec = get_decoder(cfg)                 # parameters keep requires_grad=True: the reference never freezes them
    opt = Optimizer(ref_dec, cfg)
    tb_dec = TB.build_decoder(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), fixtures.SPECS)
    prm = O.GNParams()
    obj = synth.make_object(args.seed, n_surface=2000, n_background=500)

    def run_ref():
        with contextlib.redirect_stdout(io.StringIO()):
            return opt.reconstruct_object(obj["t_cam_obj_init"].copy(), obj["pts"].copy(), obj["rays"].copy(), obj["depth"].copy())

    def run_tb():
        return TB.reconstruct_object(tb_dec, prm, obj["t_cam_obj_init"], obj["pts"], obj["rays"], obj["depth"])

    run_ref(); run_tb()     # warm-up (allocator, thread pool)
    t_ref, t_tb = [], []
    for _ in range(args.reps):
        t0 = time.perf_counter(); r = run_ref(); t_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); b = run_tb(); t_tb.append(time.perf_counter() - t0)
    d_t = float(np.abs(np.asarray(r.t_cam_obj) - b["t_cam_obj"]).max())
    d_c = float(np.abs(np.asarray(r.code) - b["code"]).max())
    m_ref, m_tb = statistics.median(t_ref), statistics.median(t_tb)
    rec = {"threads": ncpu, "reps": args.reps, "object": "cfg2 (2000 surface + 500 background rays, 10 GN iterations), seed %d" % args.seed,
           "reference_s": [round(x, 3) for x in t_ref], "torch_baseline_s": [round(x, 3) for x in t_tb],
           "reference_median_s": round(m_ref, 3), "torch_baseline_median_s": round(m_tb, 3),
           "torch_baseline_over_reference": round(m_tb / m_ref, 4), "result_max_abs_diff_pose": d_t, "result_max_abs_diff_code": d_c,
           "torch": torch.__version__}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "cpu_baseline_calibration.json"), "w") as f:
        json.dump(rec, f, indent=1)
    with open(os.path.join(ROOT, "profiles", "r04_cpu_baseline_calibration.md"), "w") as f:
        f.write("# CPU baseline calibration (round 4: five alternating repetitions)\n\n`python tools/calibrate_cpu_baseline.py` in the build container (%d host threads, torch %s).\n\n" % (ncpu, torch.__version__))
        f.write("One cfg2 object, all 10 Gauss-Newton iterations, alternating runs:\n\n| implementation | runs (s) | median (s) |\n|---|---|---|\n")
        f.write("| unmodified reference (`reconstruct/optimizer.py` via `oracle/ref_shim.py`, parameters requiring grad) | %s | %.3f |\n" % (
            ", ".join("%.2f" % x for x in t_ref), m_ref))
        f.write("| `oracle/torch_baseline.py` (what `bench.py` times on the GPU box) | %s | %.3f |\n\n" % (", ".join("%.2f" % x for x in t_tb), m_tb))
        f.write("ratio torch_baseline / reference = **%.3f**; results differ by %.2e (pose, max abs) / %.2e (code).\n" % (m_tb / m_ref, d_t, d_c))
    print(json.dumps(rec, indent=1))
    assert 0.9 <= m_tb / m_ref <= 1.1, "the restatement's time is not within 10 % of the reference's"


if __name__ == "__main__":
    main()
