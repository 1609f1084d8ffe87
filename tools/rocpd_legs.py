#!/usr/bin/env python3
"""Match a rocprofv3 kernel trace (ROCm 7.2 rocpd sqlite) of `bench.py` against the JSON line the same process printed, LEG BY LEG.

    python tools/rocpd_legs.py <results.db> <file holding the bench output> [out.md]

bench.py opens every leg of its run (headline warm-up / timed, prepass-off warm-up / timed, clock probe, latency probes, ...) with one launch
of a marker kernel nothing else runs (`k_debug_lie`), and reports per leg the launch count and HIP-event average of the decoder kernels
(`roofline.rocprof_check.legs`).  This tool splits the trace's dispatches at the marker dispatches and prints, per leg and decoder kernel,
calls / average / min / max from the TRACE next to the JSON figure and their difference -- so that each population of a kernel (one launch
per iteration of 64 objects in the headline leg, ten shorter ones in the prepass-off leg, sub-millisecond ones in the one-object legs) is
compared with its own number instead of an average over all of them (a mixed average produced a "fraction of peak" above 1).
"""
import json
import sqlite3
import sys

SHORT = (("mlp_kernelILi0", "mlp_kernel<0>"), ("mlp_kernelILi1", "mlp_kernel<1>"), ("mlp_kernelILi2", "mlp_kernel<2>"), ("mlp_kernelILi3", "mlp_kernel<3>"),
         ("mlp_lp_kernelILb0", "mlp_lp_kernel<f16>"), ("mlp_lp_kernelILb1", "mlp_lp_kernel<bf16>"), ("mlp_split_kernelILi0", "mlp_split_kernel<0>"),
         ("mlp_split_kernelILi1", "mlp_split_kernel<1>"), ("mlp_split_kernelILi2", "mlp_split_kernel<2>"), ("mlp_cluster_kernel", "mlp_cluster_kernel"),
         ("mlp_lpj_fwd_kernelILb0", "mlp_lpj_fwd_kernel<f16>"), ("mlp_lpj_bwd_kernelILb0", "mlp_lpj_bwd_kernel<f16>"),
         ("mlp_lpj_fwd_kernelILb1", "mlp_lpj_fwd_kernel<bf16>"), ("mlp_lpj_bwd_kernelILb1", "mlp_lpj_bwd_kernel<bf16>"))
JSON_KEY = {"mlp_kernel<1>": "fwd_fp32", "mlp_lp_kernel<f16>": "prepass", "mlp_lp_kernel<bf16>": "prepass"}
F_FWD = 3671040.0


def short(name):
    for key, s in SHORT:
        if key in name:
            return s
    return None


def main():
    db = sqlite3.connect(sys.argv[1])
    line = [ln for ln in open(sys.argv[2]).read().splitlines() if ln.startswith('{"metric"')][-1]
    res = json.loads(line)
    legs = res["roofline"]["rocprof_check"]["legs"]
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                      "order by d.start").fetchall()
    leg = 0
    per = {}       # (leg, kernel) -> durations
    n_mark = 0
    for name, t0, t1 in rows:
        if "k_debug_lie" in name:
            n_mark += 1
            leg = n_mark
            continue
        s = short(name)
        if s:
            per.setdefault((leg, s), []).append((t1 - t0) / 1e6)
    out = ["| leg | kernel | calls (trace) | avg ms (trace) | min ms | max ms | launches (bench JSON) | avg ms (bench JSON, HIP events) | difference |",
           "|---|---|---|---|---|---|---|---|---|"]
    worst = 0.0
    for (lg, k), d in sorted(per.items()):
        name = "before the first marker (dsp_create: prepass calibration)" if lg == 0 else (legs[lg - 1]["leg"] if lg - 1 < len(legs) else "leg %d" % lg)
        js = legs[lg - 1].get(JSON_KEY.get(k, ""), None) if 0 < lg <= len(legs) else None
        avg = sum(d) / len(d)
        jl = js["launches"] if js else None
        ja = js["avg_ms"] if js else None
        diff = ""
        if ja:
            diff = "%.2f %%" % (100.0 * (avg - ja) / ja)
            if "timed" in name:
                worst = max(worst, abs(avg - ja) / ja)
        out.append("| %s | %s | %d | %.4f | %.4f | %.4f | %s | %s | %s |" % (name, k, len(d), avg, min(d), max(d), jl if jl is not None else "-",
                                                                           "%.4f" % ja if ja else "-", diff))
    if n_mark != len(legs):
        out.append("")
        out.append("WARNING: %d marker dispatches in the trace, %d legs in the JSON line" % (n_mark, len(legs)))
    # the headline fraction recomputed from the TRACE alone
    r = res["roofline"]
    for lg, l in enumerate(legs, 1):
        if l["leg"] == "headline_timed" and (lg, "mlp_kernel<1>") in per:
            d = per[(lg, "mlp_kernel<1>")]
            avg = sum(d) / len(d)
            tf = r["alg_flop_per_launch"] / (avg * 1e-3) / 1e12
            out += ["", "Headline leg, `mlp_kernel<1>`: %d launches, trace average %.4f ms; algorithmic %.4g FLOP per launch (bench JSON: points decoded x %d) "
                        "-> **%.2f TFLOP/s = %.4f of the %.1f TFLOP/s fp32 MFMA peak** (bench JSON, HIP events: %.4f)." % (
                            len(d), avg, r["alg_flop_per_launch"], int(F_FWD), tf, tf / r["peak"], r["peak"], r["frac"])]
        if l["leg"] == "prepass_off_timed" and (lg, "mlp_kernel<1>") in per and "prepass_off" in res:
            d = per[(lg, "mlp_kernel<1>")]
            out += ["Prepass-off leg, `mlp_kernel<1>`: %d launches, trace average %.4f ms (bench JSON: %.4f ms; fraction of peak %.4f)." % (
                len(d), sum(d) / len(d), res["prepass_off"]["fwd_avg_launch_ms"], res["prepass_off"]["roofline_frac"])]
        if l["leg"] == "headline_timed" and "prepass" in res:
            for k in ("mlp_lp_kernel<f16>", "mlp_lp_kernel<bf16>"):
                if (lg, k) in per:
                    d = per[(lg, k)]
                    avg = sum(d) / len(d)
                    tf = res["prepass"]["alg_flop_per_launch"] / (avg * 1e-3) / 1e12
                    out += ["Headline leg, `%s`: %d launches, trace average %.4f ms -> %.1f TFLOP/s = %.4f of the %.0f TFLOP/s dense 16-bit peak "
                            "(bench JSON: %.4f)." % (k, len(d), avg, tf, tf / res["prepass"]["peak"], res["prepass"]["peak"], res["prepass"]["frac"])]
        if l["leg"] == "lp_compute_timed" and "lp_compute" in res:
            lp = res["lp_compute"]["roofline"]
            f, b2 = per.get((lg, "mlp_lpj_fwd_kernel<f16>")), per.get((lg, "mlp_lpj_bwd_kernel<f16>"))
            if f and b2:
                pair = sum(f) / len(f) + sum(b2) / len(b2)
                tf = lp["jacobian"]["alg_flop_per_launch_pair"] / (pair * 1e-3) / 1e12
                out += ["Low-precision compute leg, jacobian pair: `mlp_lpj_fwd_kernel<f16>` %d launches, trace average %.4f ms + `mlp_lpj_bwd_kernel<f16>` %d launches, %.4f ms = %.4f ms per pair "
                        "(bench JSON, HIP events around each: %.4f) -> %.1f TFLOP/s = %.4f of the %.0f TFLOP/s dense 16-bit peak (bench JSON: %.4f); forward share of the pair %.3f." % (
                            len(f), sum(f) / len(f), len(b2), sum(b2) / len(b2), pair, lp["jacobian"]["avg_launch_pair_ms"], tf, tf / lp["peak"], lp["peak"], lp["jacobian"]["frac"],
                            (sum(f) / len(f)) / pair)]
            k0 = per.get((lg, "mlp_lp_kernel<f16>"))
            if k0:
                out += ["Low-precision compute leg, `mlp_lp_kernel<f16>` (ray samples): %d launches, trace average %.4f ms (bench JSON: %.4f)." % (
                    len(k0), sum(k0) / len(k0), lp["ray_samples"]["avg_launch_ms"])]
    out += ["", "Largest |trace - JSON| / JSON over the timed legs: %.2f %%" % (100 * worst)]
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")


if __name__ == "__main__":
    main()
