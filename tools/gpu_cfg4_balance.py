#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU: the 1024-object job cut into 8 shards, each shard timed as the batch one GPU of an 8-GPU node would run
(VERDICT r4 item 8).  A multi-GPU step lasts as long as its slowest shard: max / mean of the shard times is what strong scaling loses to
imbalance before any hardware difference between GPUs.

    python tools/gpu_cfg4_balance.py [n_objects 1024] [n_shards 8] > profiles/r05_cfg4_balance.md

Two partitions of the same object list: the static cost of rounds 1-4 (R*D + 2M: identical for every cfg2 object -> equal counts) and the
measured cost (distributed.measure_costs: V, band and K of ONE Gauss-Newton iteration per object -> distributed.shard_objects)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsp_slam_amd import fixtures, synth, engine as E, distributed as D  # noqa: E402
from dsp_slam_amd.deep_sdf.deep_sdf_decoder import fold_weight_norm  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    layers = fold_weight_norm(fixtures.load_decoder_npz(fixtures.fixture_path("cars")), 9)
    eng = E.Engine(layers, [4], 64, device=0)
    prm = E.gn_params()
    t0 = time.time()
    objs = [synth.make_object(1 + i, n_surface=2000, n_background=500) for i in range(n)]
    t_gen = time.time() - t0
    t0 = time.time()
    costs = D.measure_costs(eng, prm, objs)
    t_meas = time.time() - t0
    static = D.shard_objects([D.object_cost(2000, 2500)] * n, world)
    measured = D.shard_objects(costs, world)

    def run(shards):
        out = []
        for a, b in shards:
            ol = objs[a:b]
            bt = eng.batch(prm, [o["t_cam_obj_init"] for o in ol], [o["pts"] for o in ol], [o["rays"] for o in ol], [o["depth"] for o in ol])
            bt.set_kernel_timing(1)
            bt.run()
            ts = []
            for _ in range(2):
                t1 = time.perf_counter()
                bt.run()
                ts.append(time.perf_counter() - t1)
            st = bt.stats()
            out.append(dict(n=b - a, ms=1e3 * min(ts), sum_V=st["n_insphere_points"], sum_K=st["n_render_rows"], fwd=st["ms_mlp_fwd"], jac=st["ms_mlp_jac"],
                            pre=st["ms_mlp_prepass"], good=int((bt.results()[3] == 0).sum())))
            bt.close()
        return out

    print("# Round 5 -- cfg4 (1024 cfg2 objects over 8 GPUs) shard balance, measured on ONE MI355X")
    print()
    print("`python tools/gpu_cfg4_balance.py %d %d`: the job's %d objects (seeds 1 .. %d) cut into %d contiguous shards; every shard run as the resident batch one GPU" % (n, world, n, n, world))
    print("of the node would hold (best of two timed runs after a warm-up, host wall clock around `dsp_batch_run`).  Object generation %.0f s (host numpy)," % t_gen)
    print("cost measurement (one Gauss-Newton iteration over all %d objects, traces read back) %.1f s = what the partitioner costs once per job." % (n, t_meas))
    print()
    for title, shards in (("static cost `R*D + 2M` (rounds 1-4): equal counts", static), ("measured first-iteration cost (`distributed.measure_costs`, round 5)", measured)):
        res = run(shards)
        ms = np.array([r["ms"] for r in res])
        print("## %s" % title)
        print()
        print("| shard | objects | good | step ms | sum V (1e6) | sum K (1e6) | fp32 forward ms | jacobian ms | prepass ms |")
        print("|---|---|---|---|---|---|---|---|---|")
        for i, r in enumerate(res):
            print("| %d | %d | %d | %.1f | %.2f | %.3f | %.1f | %.1f | %.1f |" % (i, r["n"], r["good"], r["ms"], r["sum_V"] / 1e6, r["sum_K"] / 1e6, r["fwd"], r["jac"], r["pre"]))
        print()
        print("slowest / mean = **%.4f** (slowest %.1f ms, mean %.1f ms): an 8-GPU step would run at %.1f %% of perfect strong scaling; job rate %.1f objects/s on 8 such GPUs"
              % (ms.max() / ms.mean(), ms.max(), ms.mean(), 100 * ms.mean() / ms.max(), n / (ms.max() * 1e-3)))
        print()
    eng.close()


if __name__ == "__main__":
    main()
