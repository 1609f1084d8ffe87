#!/usr/bin/env python3
"""Re-optimise every object of a saved DSP-SLAM map on the MI355X(s): the 1024-object job of BASELINE configs[3].

DSP-SLAM dumps its map at exit as MapObjects.txt -- id / 3x4 Sim(3) object->world pose / shape code per object
(src/System_util.cc:123-146; re-read by extract_map_objects.py:46-63).  The dump holds the RESULT of the per-detection optimisation, not
its inputs, so this tool takes the detections from a sidecar directory the caller fills while SLAM runs (one file per object, holding what
LocalMapping hands to Optimizer.reconstruct_object, src/LocalMapping_util.cc:179-180):

    <map_dir>/observations/<id>.npz :  pts (M,3) surface points, rays (R,3) ray directions (foreground rows first), depth (n_fg,)
                                       observed depths, all in the frame of the observing camera; t_world_cam (4,4) that camera's pose

Every object with an observation file is optimised jointly (shape code + Sim(3) pose) as ONE ragged batch per GPU, warm-started from the
saved code and pose: objects are block-sharded over the GPUs by estimated cost (dsp_slam_amd.distributed.shard_objects), each shard runs
as one dsp_batch on its own handle from its own host thread, and the result rows are gathered once.  Objects whose optimisation fails
(is_good False) or that have no observation keep their saved pose and code.  The map is written back in the reference's format.

    python tools/reoptimise_map.py --config configs/config_kitti.json --map_dir map/kitti/07 [--gpus N] [--out MapObjects.reopt.txt]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dsp_slam_amd"))
sys.path.insert(0, ROOT)


def load_observations(map_dir, objs):
    """-> list parallel to objs: dict(pts, rays, depth, t_world_cam) or None."""
    out = []
    for o in objs:
        p = os.path.join(map_dir, "observations", "%d.npz" % o["id"])
        if not os.path.exists(p):
            out.append(None)
            continue
        z = np.load(p)
        out.append(dict(pts=np.ascontiguousarray(z["pts"], np.float32), rays=np.ascontiguousarray(z["rays"], np.float32),
                        depth=np.ascontiguousarray(z["depth"], np.float32).reshape(-1), t_world_cam=np.asarray(z["t_world_cam"], np.float64)))
    return out


def reoptimise(engines, prm, objs, obs, code_len, shards=None, compute=0):
    """objs / obs as read; engines: one dsp_slam_amd.engine.Engine per GPU.  -> (objects with updated pose / code, stats dict).
    shards: optional explicit (start, stop) blocks over the objects that have observations (default: cost-balanced over the engines).
    compute: 0 = fp32 (the parity path), 1 / 2 = the opt-in f16 / bf16 compute mode (include/dsp_gn.h: dsp_batch_set_compute)."""
    from dsp_slam_amd import distributed as D
    idx = [i for i, ob in enumerate(obs) if ob is not None]
    t_in, codes_in = [], []
    for i in idx:
        t_wc = obs[i]["t_world_cam"]
        t_in.append((np.linalg.inv(t_wc) @ np.asarray(objs[i]["pose"], np.float64)).astype(np.float32))     # object -> camera, the optimiser's frame
        codes_in.append(np.asarray(objs[i]["code"], np.float32)[:code_len])
    if shards is None:
        shards = D.shard_objects([D.object_cost(obs[i]["pts"].shape[0], obs[i]["rays"].shape[0], prm.num_depth_samples) for i in idx], len(engines))
    parts = [None] * len(shards)

    def work(r):
        a, b = shards[r]
        sel = idx[a:b]
        eng = engines[r % len(engines)]
        res = eng.reconstruct_batch(prm, t_in[a:b], [obs[i]["pts"] for i in sel], [obs[i]["rays"] for i in sel], [obs[i]["depth"] for i in sel],
                                    codes_in[a:b], compute=compute)
        parts[r] = D.pack_results(*res)

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(len(shards))]      # ctypes releases the GIL: one host thread per GPU
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    packed = np.concatenate([p for p in parts if p is not None and p.shape[0]], 0) if idx else np.zeros((0, D.RESULT_WIDTH), np.float32)
    t, codes, loss, status = D.unpack_results(packed)
    out = [dict(o) for o in objs]
    n_good = 0
    for k, i in enumerate(idx):
        if status[k] != 0:
            continue
        n_good += 1
        out[i]["pose"] = obs[i]["t_world_cam"] @ t[k].astype(np.float64)
        out[i]["code"] = codes[k, :len(objs[i]["code"])].astype(np.float32)
        out[i]["loss"] = float(loss[k])
    return out, dict(n_objects=len(objs), n_observed=len(idx), n_good=n_good, seconds=dt, shards=[tuple(s) for s in shards], packed=packed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--map_dir", required=True)
    ap.add_argument("--gpus", type=int, default=0, help="0 = every MI355X the library accepts")
    ap.add_argument("--out", default=None, help="default: <map_dir>/MapObjects.reopt.txt")
    ap.add_argument("--compute", choices=("f32", "f16", "bf16"), default="f32",
                    help="f32 = the parity path (default); f16 / bf16 = the opt-in low-precision compute mode: ~3.7 x the objects/s on large maps, accuracy in profiles/r06_lp_compute.md")
    args = ap.parse_args()
    from reconstruct.utils import get_configs
    from deep_sdf.workspace import config_decoder
    from dsp_slam_amd import _lib as L, engine as E
    from dsp_slam_amd.map_objects import read_map_objects, write_map_objects
    cfg = get_configs(args.config)
    objs = read_map_objects(os.path.join(args.map_dir, "MapObjects.txt"))
    obs = load_observations(args.map_dir, objs)
    n_dev = args.gpus or max(1, L.load().dsp_device_count())
    decoders = [config_decoder(cfg.DeepSDF_DIR).cuda(d) for d in range(n_dev)]        # one decoder (= one handle, one stream) per GPU
    prm = E.params_from_configs(cfg)
    out, st = reoptimise([d.engine for d in decoders], prm, objs, obs, cfg.optimizer.code_len, compute={"f32": 0, "f16": 1, "bf16": 2}[args.compute])
    dst = args.out or os.path.join(args.map_dir, "MapObjects.reopt.txt")
    write_map_objects(dst, out)
    print("re-optimised %d of %d objects (%d with observations) on %d GPU(s) in %.3f s = %.1f objects/s -> %s" % (
        st["n_good"], st["n_objects"], st["n_observed"], n_dev, st["seconds"], st["n_observed"] / max(st["seconds"], 1e-9), dst))


if __name__ == "__main__":
    main()
