#!/usr/bin/env python3
"""Development aid (GPU box): per-detection latency A/B of the round-4 switches, host wall clock p50 around run + results of a resident
one-object batch, plus the one-shot entry point.  python tools/gpu_latency_ab.py [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
eng = E.Engine(fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9), [4], 64, device=0)
prm = E.gn_params()


def p50(fn):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


for name, (M, Bg) in (("KITTI-size 250+200", (250, 200)), ("cfg2-size 2000+500", (2000, 500))):
    o = synth.make_object(4242 if M == 250 else 1, n_surface=M, n_background=Bg)
    args = ([o["t_cam_obj_init"]], [o["pts"]], [o["rays"]], [o["depth"]])
    print("## %s" % name)
    variants = [("automatic (cluster tiles, wave bookkeeping, rows-in-lanes solve, no events)    ", {}),
                ("one workgroup per jacobian tile (cluster form off)", dict(cluster_tiles=0)),
                ("render rows repeat their forward sweep (mixed mask reuse off)", dict(mixed_reuse=0)),
                ("throughput bookkeeping", dict(wave_bookkeeping=0)),
                ("per-kernel events on", dict(kernel_timing=1)),
                ("no speculative band rows", dict(speculative_band=0)),
                ("tile lists built by k_build_tiles (direct tiles off)", dict(direct_tiles=0)),
                # (the per-object fused bookkeeping and the earlier solver kernels were measured in rounds 3-5 -- profiles/r04_latency_ab.md,
                #  profiles/r05_latency_ab.md -- and left the library in round 6: profiles/r06_removed_experiments.md)
                ("prepass off", dict(prepass=0)),
                ("OPT-IN low-precision compute mode, f16 (NOT the parity path; a detection-sized batch keeps the fp32 path)", dict(compute=1)),
                ("... pinned on for this batch whatever its size (DSP_DBG_LP_SMALL_BATCHES)", dict(compute=1, lp_small_batches=1))]
    ref = None
    for label, kw in variants:
        b = eng.batch(prm, *args)
        for k, v in kw.items():
            getattr(b, "set_" + k)(v)

        def run():
            b.run()
            b.results()
        med, mn = p50(run)
        st = b.stats()
        res = b.results()
        if ref is None:
            ref = res
        same = all(np.array_equal(x, y) for x, y in zip(res, ref))
        print("%-78s p50 %7.3f ms  min %7.3f  device %7.3f ms  launches fwd/jac/lp %d/%d/%d  cluster tiles %d  bits==auto %s" % (
            label, med, mn, st["ms_total"], st["n_mlp_fwd_launches"], st["n_mlp_jac_launches"], st["n_mlp_prepass_launches"], st["n_cluster_tiles"], same), flush=True)
        b.close()
    med, mn = p50(lambda: eng.reconstruct_batch(prm, *args))
    print("%-70s p50 %7.3f ms  min %7.3f" % ("one-shot dsp_reconstruct_batch (host buffers in, results out)", med, mn), flush=True)
eng.close()
