#!/usr/bin/env python3
"""Benchmark of the DeepSDF shape/pose Gauss-Newton hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (Optimizer.reconstruct_object, all 10 GN iterations) over one batch of
synthetic cfg2 objects (2000 surface points + 500 background rays x 50 depth samples, 64-D code) per GPU, inputs
already resident in HBM (dsp_batch_create uploads them before the timed region), followed by one RCCL gather
of the per-object results (pose 16 + code 64 + loss + status floats) to rank 0.  Objects are independent, so
ranks shard them with no other collective: weak scaling, value = objects of all ranks / max-over-ranks time.

The JSON line also carries
  roofline      fp32-MFMA roofline of the dominant kernel (forward-only decoder, mlp_kernel<false>): algorithmic
                FLOPs = points the launches actually decode x 3 671 040 FLOP / HIP-event time of those
                launches.  The path decodes fewer points than the reference's V in-sphere samples: samples behind the
                first solid sample of a ray have exactly zero transmittance and are skipped (results identical);
  cpu_baseline  the CPU oracle (oracle/dsp_oracle.py, torch-CPU sgemm) timed on this box's host cores on ONE cfg2
                object (rank 0, N=1 only) -- a reported baseline, not a target.
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_FWD = 3671040.0          # FLOP / point, decoder forward                
F_JAC = 7342080.0          # FLOP / point, forward + input-gradient
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--objects-per-gpu", type=int, default=64)   # BASELINE configs[2]: batches of 64 cfg2 objects saturate the matrix pipe
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-runs", type=int, default=9)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs: python -m torch.distributed.run --nproc-per-node %d bench.py ..." % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("DSP_BENCH_FORCE_DIST") == "1":   # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from dsp_slam_amd import fixtures, synth, engine as E
    from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm

    sd = fixtures.load_decoder_npz(fixtures.fixture_path("cars"))
    layers = fold_weight_norm(sd, len(fixtures.SPECS["NetworkSpecs"]["dims"]) + 1)
    eng = E.Engine(layers, fixtures.SPECS["NetworkSpecs"]["latent_in"], fixtures.SPECS["CodeLength"], device=local_rank)
    prm = E.gn_params()      # KITTI hyper-parameters (configs/config_kitti.json:21-40 of the reference)
    B = args.objects_per_gpu
    objs = synth.make_batch(B, first_seed=1 + rank * B, n_surface=2000, n_background=500)
    batch = eng.batch(prm, [o["t_cam_obj_init"] for o in objs], [o["pts"] for o in objs], [o["rays"] for o in objs],
                      [o["depth"] for o in objs])

    gathered = [None]

    from dsp_slam_amd import distributed as D
    shards = [(r * B, (r + 1) * B) for r in range(world)]     # weak scaling: B objects on every rank

    def step():
        batch.run()
        packed = D.pack_results(*batch.results())
        if dist is not None:     # the single collective of the path: results to rank 0 over RCCL / xGMI
            gathered[0] = D.gather_results(packed, shards, dist, device=torch.device("cuda", local_rank))
        else:
            gathered[0] = packed

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    fwd_ms = jac_ms = fwd_pts = jac_pts = ren_rows = insphere_pts = 0.0
    n_fwd = n_jac = 0
    for _ in range(args.steps):
        step()
        st = batch.stats()
        fwd_ms += st["ms_mlp_fwd"]; jac_ms += st["ms_mlp_jac"]
        fwd_pts += st["n_fwd_points"]; jac_pts += st["n_jac_points"]; ren_rows += st["n_render_rows"]; insphere_pts += st["n_insphere_points"]
        n_fwd += st["n_mlp_fwd_launches"]; n_jac += st["n_mlp_jac_launches"]
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    n_good = int((batch.results()[3] == 0).sum())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = world * B * args.steps / elapsed
    # fabric bytes per decoded point from the last committed rocprofv3 --pmc pass (tools/rocpd_pmc.py, FETCH_SIZE x 2 on gfx950)
    pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
    fwd_tflops = fwd_pts * F_FWD / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
    # surface points run forward + backward; render rows only the backward sweep (masks come from the forward launches)
    jac_flop = jac_pts * F_JAC + ren_rows * (F_JAC - F_FWD)
    jac_tflops = jac_flop / (jac_ms * 1e-3) / 1e12 if jac_ms > 0 else 0.0
    result = {
        "metric": "objects/sec (2000 pts, 64-D code, 10 GN iters)",
        "value": round(value, 3),
        "unit": "objects/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded rounded-box objects, decoder fixture fitted to them; no real weights/datasets offline)",
        "config": {
            "workload": "cfg2: single KITTI-like car per object -- 2000 surface pts + 500 free-space rays (2500 rays x 50 depth "
                        "samples), 64-D code, 10 joint GN iterations (Optimizer.reconstruct_object), batch of %d objects per GPU" % B,
            "objects_per_gpu": B,
            "objects_good": n_good,
            "parallelism": "object-sharded x%d, one RCCL gather of results per step" % world,
        },
        "roofline": {
            "bound": "mfma",
            "kernel": "mlp_kernel<1,false> (decoder forward with relu-mask export, fp32 v_mfma_f32_16x16x4_f32)",
            "achieved": round(fwd_tflops, 2),
            "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(fwd_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
            "traffic": pmc.get("fwd_fetch_bytes_per_point", 0.0) * fwd_pts / max(n_fwd, 1) or None,
            "traffic_note": pmc.get("note", "no PMC pass recorded (profiles/pmc_traffic.json missing)"),
            "avg_launch_ms": round(fwd_ms / max(n_fwd, 1), 4),
            "alg_flop_per_launch": round(fwd_pts * F_FWD / max(n_fwd, 1)),
            "fwd_points_evaluated_over_insphere": round(fwd_pts / max(insphere_pts, 1.0), 4),
            "jac_kernel_tflops": round(jac_tflops, 2),
            "jac_kernel_frac": round(jac_tflops / PEAK_FP32_MFMA_TFLOPS, 4),
            "jac_avg_launch_ms": round(jac_ms / max(n_jac, 1), 4),
            "whole_path_tflops": round((fwd_pts * F_FWD + jac_flop) / elapsed / 1e12 * world, 2),
        },
    }

    if world == 1:
        # single-object latency (ms/object p50): batch of ONE cfg2 object
        o = objs[0]
        one = eng.batch(prm, [o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
        one.run()
        lat = []
        for _ in range(max(args.latency_runs, 1)):
            t1 = time.perf_counter()
            one.run()
            one.results()
            lat.append((time.perf_counter() - t1) * 1e3)
        one.close()
        result["latency_ms_p50"] = round(statistics.median(lat), 3)
        # sdf-only workload: Optimizer.estimate_pose_cam_obj on the same objects -- 5 Gauss-Newton iterations of the
        # surface term alone, through the one-shot entry point (host buffers in, so this figure includes upload and download)
        try:
            t_se3, scales = [], []
            for o in objs:
                t = np.array(o["t_cam_obj_init"], np.float32)
                sc = float(np.cbrt(np.linalg.det(t[:3, :3].astype(np.float64))))
                t[:3, :3] /= sc
                t_se3.append(t); scales.append(sc)
            zero_codes = [np.zeros(64, np.float32)] * len(objs)
            pts_list = [o["pts"] for o in objs]
            eng.estimate_pose_batch(prm, t_se3, scales, pts_list, zero_codes)
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                eng.estimate_pose_batch(prm, t_se3, scales, pts_list, zero_codes)
            dt_pose = (time.perf_counter() - t1) / reps
            result["pose_only"] = {"objects_per_s": round(len(objs) / dt_pose, 1), "ms_per_batch": round(dt_pose * 1e3, 3),
                                   "note": "estimate_pose_cam_obj, %d objects x 2000 points x 5 iterations per call, host buffers in/out" % len(objs)}
        except Exception as e:      # never lose the headline line over the secondary figure
            result["pose_only"] = {"error": repr(e)}

    if world == 1 and not args.no_cpu_baseline:
        from oracle import dsp_oracle as O      # checker/baseline only -- never on the product path
        dec = O.fold_decoder(sd, fixtures.SPECS)
        oprm = O.GNParams()
        # pick the intra-op thread count that runs a small object fastest (128-thread hosts are slower at 128 than at 16-32)
        small = synth.make_object(999, n_surface=500, n_background=0)
        oprm5 = O.GNParams(num_iterations=2)
        ncpu = os.cpu_count() or 1
        best_threads, best_t = None, None
        for nt in sorted({min(ncpu, x) for x in (8, 16, 32, 64, ncpu)}):
            torch.set_num_threads(nt)
            O.reconstruct_object(dec, oprm5, small["t_cam_obj_init"], small["pts"], small["rays"], small["depth"])
            t1 = time.perf_counter()
            O.reconstruct_object(dec, oprm5, small["t_cam_obj_init"], small["pts"], small["rays"], small["depth"])
            dt_s = time.perf_counter() - t1
            if best_t is None or dt_s < best_t:
                best_threads, best_t = nt, dt_s
        torch.set_num_threads(best_threads)
        o = objs[0]
        t1 = time.perf_counter()
        r = O.reconstruct_object(dec, oprm, o["t_cam_obj_init"], o["pts"], o["rays"], o["depth"])
        dt = time.perf_counter() - t1
        gpu_t = D.unpack_results(gathered[0])[0][0]
        result["cpu_baseline"] = {
            "value": round(1.0 / dt, 4),
            "unit": "objects/s",
            "cores": int(best_threads),
            "kind": "port",
            "sample": "1 cfg2 object (seed %d), all 10 GN iterations, oracle/dsp_oracle.py with torch-CPU sgemm on %d of %d host threads "
                      "(fastest of a small sweep); %.2f s" % (1 + rank * B, best_threads, ncpu, dt),
            "gpu_vs_cpu": round(value * dt, 1),
            "pose_max_abs_diff_vs_gpu": float(np.abs(r["t_cam_obj"] - gpu_t).max()) if r["is_good"] else None,
        }
    batch.close()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    try:    # RCCL prints its banner through C stdio; flush it so that the JSON is the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
