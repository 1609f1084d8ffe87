#!/usr/bin/env python3
"""profiles/<tag>_*.md and profiles/pmc_traffic.json from the outputs of tools/run_profiles.sh (gpurun_out/<tag>/), with the
derived numbers (TFLOP/s, matrix-pipe busy, clocks, bytes per point) computed here rather than by hand.

    python tools/make_profiles.py [gpurun_out/r03 [tag]]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03")
TAG = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.path.normpath(SRC))     # file prefix under profiles/ (r03, r03b, ...)
ROUND = TAG[1:3].lstrip("0") or "?"
DST = os.path.join(ROOT, "profiles")
F_FWD = 3671040.0
PEAK32, PEAK16 = 157.3, 2500.0

K1 = "_ZN3dsp10mlp_kernelILi1EEEvNS_7MlpArgsE.kd"
K2 = "_ZN3dsp10mlp_kernelILi2EEEvNS_7MlpArgsE.kd"
K2R = "_ZN3dsp10mlp_kernelILi3EEEvNS_7MlpArgsE.kd"
K0 = "_ZN3dsp13mlp_lp_kernelILb0ELi2EEEvNS_6LpArgsE.kd"      # f16, two column blocks per wave: the 128-point-tile form
KJF = "_ZN3dsp18mlp_lpj_fwd_kernelILb0EEEvNS_7LpjArgsE.kd"   # low-precision compute mode: forward with mask export
KJB = "_ZN3dsp18mlp_lpj_bwd_kernelILb0EEEvNS_7LpjArgsE.kd"   # ... backward from the masks


def read(name):
    return open(os.path.join(SRC, name)).read()


def bench(name):
    return json.loads([ln for ln in read(name).splitlines() if ln.startswith('{"metric"')][-1])


def pmc(name):
    rows, dur = {}, {}
    for line in read(name).splitlines():
        m = re.match(r"\| (\S+) \| (\S+) \| (\d+) \| (\S+) \| (\S+) \|", line)
        if m and m.group(2) != "counter":
            rows[(m.group(1), m.group(2))] = (int(m.group(3)), float(m.group(4)))
        m = re.match(r"duration: (\S+)\s+dispatches (\d+)\s+total ([\d.]+) ms", line)
        if m:
            dur[m.group(1)] = (int(m.group(2)), float(m.group(3)))
    return rows, dur


def stats_rows(name):
    out = {}
    for line in read(name).splitlines():
        c = [x.strip() for x in line.split("|")]
        if len(c) > 8 and c[1].startswith("_ZN3dsp") or (len(c) > 8 and c[1].startswith("__amd")):
            out[c[1]] = dict(calls=int(c[2]), total_ms=float(c[3]), avg_us=float(c[4]), pct=float(c[7]))
    return out


def table_only(name):
    return "\n".join(l for l in read(name).splitlines() if l.startswith("|") or l.startswith("duration:"))


def main():
    b, b4, b5 = bench("bench.json"), bench("bench_cfg4.json"), bench("bench_cfg5.json")
    boff = bench("bench_prepass_off.json") if os.path.exists(os.path.join(SRC, "bench_prepass_off.json")) else None
    cmd = "python bench.py --steps %d --warmup %d" % (b["steps"], b["warmup"])

    # ---- bench lines -------------------------------------------------------------------------------------------------------------
    def row(label, d):
        r, p = d["roofline"], d.get("prepass")
        return "| %s | %.1f | %.1f | %.3f | %.3f | %s |" % (label, d["value"], d["ms_per_step"], r["frac"], r["jac_kernel_frac"],
                                                         "%.0f (%.3f)" % (p["achieved"], p["frac"]) if p else "-")
    lines = ["# Round %s -- bench lines" % ROUND + " as printed on an MI355X (one gpurun call, `tools/run_profiles.sh`; this file by `tools/make_profiles.py`)",
             "", "`%s [--config ...]` (cfg2x64: the driver's own command)" % cmd, "",
             "| config | objects/s | ms per step | fp32 forward kernel frac of 157.3 TFLOP/s | jacobian kernels frac | prepass kernel TFLOP/s (frac of 2500) |", "|---|---|---|---|---|---|",
             row("cfg2x64 (dtype `%s`)" % b["dtype"], b)]
    if "prepass_off" in b:
        po = b["prepass_off"]
        lines += ["| cfg2x64, prepass off = `value_fp32_only` (same process, %d steps) | %.1f | %.1f | %.3f | %.3f | - |" % (
            po["steps"], po["value"], po["ms_per_step"], po["roofline_frac"], po["jac_kernel_frac"])]
    if "lp_compute" in b:
        lc = b["lp_compute"]
        lines += ["| cfg2x64, OPT-IN low-precision compute mode = `value_lp` (same process, %d steps; NOT the parity path) | %.1f | %.1f | - | f16 jacobian pair %.0f TFLOP/s (%.3f of 2500) | ray samples %.0f (%.3f) |" % (
            lc["steps"], lc["value"], lc["ms_per_step"], lc["roofline"]["jacobian"]["achieved"], lc["roofline"]["jacobian"]["frac"], lc["roofline"]["ray_samples"]["achieved"],
            lc["roofline"]["ray_samples"]["frac"])]
    if boff:
        lines += [row("cfg2x64, --prepass off (own process)", boff)]
    lines += [row("cfg4 (%s scaling: %d objects on this GPU)" % (b4["scaling"], b4["config"]["objects_per_gpu"]), b4), row("cfg5", b5), ""]
    for title, d in (("cfg2x64 (the headline configuration), f16 prepass", b), ("cfg2x64, --prepass off (own process)", boff),
                     ("cfg4: the 1024-object job (strong scaling; on one GPU the whole job is one shard)", b4),
                     ("cfg5: 4000-point objects, Redwood hyper-parameters, cars + chairs32 decoders", b5)):
        if d:
            lines += ["## " + title, "", "```json", json.dumps(d), "```", ""]
    open(os.path.join(DST, TAG + "_bench_lines.md"), "w").write("\n".join(lines))

    # ---- kernel stats ------------------------------------------------------------------------------------------------------------
    # rocprofv3 --kernel-trace --stats of the DRIVER'S command, split into the legs bench.py marks (tools/rocpd_legs.py): each population of a
    # kernel -- one launch per iteration of 64 objects in the headline leg, ten shorter ones in the prepass-off leg, sub-millisecond ones
    # in the one-object legs -- is compared with its own HIP-event figure.  (Round 4 averaged them all and printed a fraction above 1.)
    r = b["roofline"]
    by = r["ms_per_step_by_kernel"]
    lp = b["prepass"]
    pts_per_launch = r["alg_flop_per_launch"] / F_FWD
    prof = bench("bench_under_rocprof_full.txt")
    text = ["# Round %s -- rocprofv3 kernel trace of the driver's bench command, leg by leg" % ROUND, "",
            "`cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -- %s`" % cmd,
            "(64 cfg2 objects per step; every leg of the process opens with one dispatch of the marker kernel `k_debug_lie`: headline warm-up / timed,",
            "prepass-off warm-up / timed, clock probe, three latency probes, pose-only).  Tables by `tools/rocpd_legs.py` and `tools/rocpd_stats.py`, this file by",
            "`tools/make_profiles.py`.  The JSON line printed by THIS profiled process: %.1f objects/s, %.1f ms per step, `roofline.frac` %.4f;" % (
                prof["value"], prof["ms_per_step"], prof["roofline"]["frac"]),
            "un-profiled, same build: `profiles/" + TAG + "_bench_lines.md` (%.1f objects/s, %.1f ms per step, `roofline.frac` %.4f)." % (b["value"], b["ms_per_step"], r["frac"]),
            "", "## Per leg: trace vs the bench line's `roofline.rocprof_check`", "", read("legs.md").rstrip(), "",
            "## Whole process, per kernel (`rocprofv3 --stats` view)", "",
            "Averages in this table run over EVERY launch of a kernel in the process -- for `mlp_kernel<1>`, `<2>`, `<3>` and `mlp_lp_kernel` that mixes the legs above",
            "(launches of 21 ms, 10 ms and < 1 ms): it is the per-leg table, not this one, that prices a kernel.", "",
            table_only("kernel_stats_full.md"), "",
            "Reading (headline leg, per step of 64 objects x 10 iterations): `mlp_kernel<1>` 10 launches of %.0f k points = %.2f TFLOP each; `mlp_kernel<2>` + `<3>`"
            % (pts_per_launch / 1e3, r["alg_flop_per_launch"] / 1e12),
            "together %.3f of the fp32 peak; `mlp_lp_kernel<f16>` 100 launches x %.2f ms = %.0f ms per step, %.2f PFLOP/s = %.3f of the 2.5 PFLOP/s dense 16-bit peak (priced"
            % (r["jac_kernel_frac"], lp["avg_launch_ms"], by["prepass"], lp["achieved"] / 1e3, lp["frac"]),
            "separately from the fp32 fraction); everything else (sampling, band selection, occupancy scan, compaction, Gram, solve, tile lists) %.1f ms per step = %.1f %%."
            % (by["other"], 100 * by["other"] / b["ms_per_step"]), ""]
    open(os.path.join(DST, TAG + "_kernel_stats.md"), "w").write("\n".join(text))
    st = stats_rows("kernel_stats_full.md")

    # ---- PMC ---------------------------------------------------------------------------------------------------------------------
    mf, dm = pmc("pmc_mfma.md")
    fe, df = pmc("pmc_fetch.md")
    wr, _ = pmc("pmc_write.md")
    ld, _ = pmc("pmc_lds.md")

    def busy(k):
        return mf[(k, "SQ_VALU_MFMA_BUSY_CYCLES")][1] / (mf[(k, "GRBM_GUI_ACTIVE")][1] / 8 * 1024)

    def clk(k):
        return mf[(k, "GRBM_GUI_ACTIVE")][1] / 8 / (dm[k][1] * 1e-3) / 1e9

    n1 = dm[K1][0]      # the PMC tables hold the headline leg only (rocpd_pmc.py --between 1 3: warm-up + timed step = 2 steps x 10 iterations)
    assert n1 % 10 == 0, n1
    # points per launch from the un-profiled bench of the same run (same workload, same seeds)
    k1_pts = pts_per_launch * n1
    k1_fetch = fe[(K1, "FETCH_SIZE")][1] * 1024 * 2
    k0_pts = lp["alg_flop_per_launch"] / F_FWD * dm[K0][0]
    k0_fetch = fe[(K0, "FETCH_SIZE")][1] * 1024 * 2
    wave1 = mf[(K1, "SQ_WAVE_CYCLES")][1]
    wave0 = ld[(K0, "SQ_ACTIVE_INST_ANY")][1] + 0.0
    txt = ["# Round %s -- rocprofv3 PMC passes" % ROUND + " at the bench configuration", "",
           "Four separate `--pmc` passes (no `--stats`, no tracing; one counter group per run) over",
           "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --latency-runs 1` -- **64 objects per GPU, the bench configuration** (round 1's pass was",
           "taken at 32).  The tables hold the HEADLINE LEG only (`tools/rocpd_pmc.py ... --between 1 3`: the dispatches between bench.py's first and third marker): 2 steps = %d launches of the fp32 forward kernel `mlp_kernel<1>`, %d of the prepass kernel." % (n1, dm[K0][0]),
           "Tables by `tools/rocpd_pmc.py` (kernels matching `mlp_`), this file by `tools/make_profiles.py`.", "",
           "## FETCH_SIZE (KiB)", "", table_only("pmc_fetch.md"), "", "## WRITE_SIZE (KiB)", "", table_only("pmc_write.md"), "",
           "## MFMA / busy counters", "", table_only("pmc_mfma.md"), "", "## LDS / wait counters", "", table_only("pmc_lds.md"), "", "## Reading", "",
           "**`mlp_kernel<1>` -- fp32 forward decoder over the unclassified band (%d launches, %.1f ms in the counter run)**" % (n1, dm[K1][1]), "",
           "* `SQ_INSTS_VALU_MFMA_MOPS_F32` %.4g x 512 FLOP = %.1f TFLOP issued to the matrix pipe = %.1f TFLOP/s (the bench prices the algorithmic"
           % (mf[(K1, "SQ_INSTS_VALU_MFMA_MOPS_F32")][1], mf[(K1, "SQ_INSTS_VALU_MFMA_MOPS_F32")][1] * 512 / 1e12,
              mf[(K1, "SQ_INSTS_VALU_MFMA_MOPS_F32")][1] * 512 / (dm[K1][1] * 1e-3) / 1e12),
           "  3.671 MFLOP/point, %.1f TFLOP/s: layer 0 and the code columns run on the VALU / as per-object biases)." % r["achieved"],
           "* Matrix pipe busy: `SQ_VALU_MFMA_BUSY_CYCLES` %.4g / (`GRBM_GUI_ACTIVE` %.4g / 8 XCDs x 256 CUs x 4 SIMDs) = **%.1f %%**"
           % (mf[(K1, "SQ_VALU_MFMA_BUSY_CYCLES")][1], mf[(K1, "GRBM_GUI_ACTIVE")][1], 100 * busy(K1)),
           "  (first round-2 profile: 84.0 %%; round 1: 80.9 %%); shader clock %.2f GHz." % clk(K1),
           "* Fabric reads: `FETCH_SIZE` %.4g KiB x 2 (gfx950 half-count correction, MI355X_MICROARCH.md) = %.1f GB over %.2f M points ="
           % (fe[(K1, "FETCH_SIZE")][1], k1_fetch / 1e9, k1_pts / 1e6),
           "  **%.2f KB per decoded point, %.2f GB per launch, %.0f GB/s = %.1f %% of the 8 TB/s HBM peak**: the 6.8 MB fp32 weight stream does not fit an"
           % (k1_fetch / k1_pts / 1e3, k1_fetch / n1 / 1e9, k1_fetch / (df[K1][1] * 1e-3) / 1e9, 100 * k1_fetch / (df[K1][1] * 1e-3) / 8e12),
           "  XCD's 4 MiB L2 and is re-fetched from the Infinity Cache by ~3 % of the 106 KB every tile streams from L2.  Algorithmic: 20 B per point.  Not a bound.",
           "* `WRITE_SIZE` %.4g KiB: sdf values and the exported relu masks of band samples (512 B each)." % wr[(K1, "WRITE_SIZE")][1],
           "* LDS: %d bank conflicts; `SQ_WAIT_INST_LDS` %.2f %% of wave cycles."
           % (ld[(K1, "SQ_LDS_BANK_CONFLICT")][1], 100 * ld[(K1, "SQ_WAIT_INST_LDS")][1] / wave1), "",
           "**`mlp_lp_kernel<f16>` -- the prepass (%d launches, %.1f ms)**" % (dm[K0][0], dm[K0][1]), "",
           "* `SQ_INSTS_VALU_MFMA_MOPS_F16` %.4g x 512 FLOP = %.0f TFLOP = **%.2f PFLOP/s issued** (algorithmic %.2f in the bench: the 445-row layer is padded to"
           % (mf[(K0, "SQ_INSTS_VALU_MFMA_MOPS_F16")][1], mf[(K0, "SQ_INSTS_VALU_MFMA_MOPS_F16")][1] * 512 / 1e12,
              mf[(K0, "SQ_INSTS_VALU_MFMA_MOPS_F16")][1] * 512 / (dm[K0][1] * 1e-3) / 1e15, lp["achieved"] / 1e3),
           "  512 rows and the xyz k-steps carry split-precision products).",
           "* Matrix pipe busy: **%.1f %%** (first round-2 profile: 53.8 %%); shader clock averaged over launches of very different length %.2f GHz (the long"
           % (100 * busy(K0), clk(K0)),
           "  ones run at 1.75-1.9 GHz: `tools/probes/gpu_prepass_probe.py`) -- the clock this chip grants a dense 16-bit MFMA stream follows the switching power of",
           "  live operands: 1.78 GHz for a register-only 32x32x16 stream, 2.13 GHz for the 16x16x32 form this kernel uses since round 5 (`profiles/r05_k0_clock.md`).",
           "* Fabric reads: %.4g KiB x 2 = %.1f GB over %.1f M points = **%.0f B per point**: the 3.6 MB f16 weight stream fits the 4 MiB L2."
           % (fe[(K0, "FETCH_SIZE")][1], k0_fetch / 1e9, k0_pts / 1e6, k0_fetch / k0_pts),
           "  `WRITE_SIZE` %.4g KiB = %.0f B per point (4 B of sdf per point + write-allocate granularity; the kernel has no scratch any more)."
           % (wr[(K0, "WRITE_SIZE")][1], wr[(K0, "WRITE_SIZE")][1] * 1024 / k0_pts),
           "* LDS: %d bank conflicts (A fragments are read as lane-linear `ds_read_b128`)." % ld[(K0, "SQ_LDS_BANK_CONFLICT")][1], "",
           "**Jacobian kernels**: `mlp_kernel<3>` (backward only) %.1f %% matrix-pipe busy, `mlp_kernel<2>` (forward + backward) %.1f %%." % (100 * busy(K2R), 100 * busy(K2)), ""]
    # ---- the low-precision compute leg (markers 3 .. 5 of the same counter runs) ----
    if os.path.exists(os.path.join(SRC, "pmc_lp_mfma.md")) and "lp_compute" in b:
        mfl, dml = pmc("pmc_lp_mfma.md")
        fel, dfl = pmc("pmc_lp_fetch.md")
        wrl, _ = pmc("pmc_lp_write.md")
        ldl, _ = pmc("pmc_lp_lds.md")
        lpc = b["lp_compute"]["roofline"]

        def busy_l(k):
            return mfl[(k, "SQ_VALU_MFMA_BUSY_CYCLES")][1] / (mfl[(k, "GRBM_GUI_ACTIVE")][1] / 8 * 1024)

        def clk_l(k):
            return mfl[(k, "GRBM_GUI_ACTIVE")][1] / 8 / (dml[k][1] * 1e-3) / 1e9

        jac_pts = lpc["jacobian"]["alg_flop_per_launch_pair"] / (2 * F_FWD) * dml[KJF][0]
        txt += ["## The low-precision compute leg (`dsp_batch_set_compute(F16)`; the same counter runs, dispatches between markers 3 and 5: 2 steps)", "",
                "### FETCH_SIZE (KiB)", "", table_only("pmc_lp_fetch.md"), "", "### WRITE_SIZE (KiB)", "", table_only("pmc_lp_write.md"), "",
                "### MFMA / busy counters", "", table_only("pmc_lp_mfma.md"), "", "### LDS / wait counters", "", table_only("pmc_lp_lds.md"), "", "### Reading", ""]
        for k, label in ((KJF, "`mlp_lpj_fwd_kernel<f16>` (forward + relu-mask export over the jacobian rows)"), (KJB, "`mlp_lpj_bwd_kernel<f16>` (backward over the transposed stream)"),
                         (K0, "`mlp_lp_kernel<f16>` (the ray samples: here its values are the results)")):
            if k not in dml:
                continue
            issued = mfl[(k, "SQ_INSTS_VALU_MFMA_MOPS_F16")][1] * 512
            fetch = fel[(k, "FETCH_SIZE")][1] * 1024 * 2
            write = wrl[(k, "WRITE_SIZE")][1] * 1024
            txt += ["* %s: %d launches, %.1f ms; `SQ_INSTS_VALU_MFMA_MOPS_F16` x 512 = %.1f TFLOP issued = **%.2f PFLOP/s**; matrix pipe busy **%.1f %%**, shader clock %.2f GHz; "
                    "fabric reads %.2f GB (%.0f GB/s), writes %.2f GB; %d LDS bank conflicts." % (
                        label, dml[k][0], dml[k][1], issued / 1e12, issued / (dml[k][1] * 1e-3) / 1e15, 100 * busy_l(k), clk_l(k), fetch / 1e9, fetch / (dfl[k][1] * 1e-3) / 1e9,
                        write / 1e9, ldl[(k, "SQ_LDS_BANK_CONFLICT")][1])]
        if KJF in dml and KJB in dml:
            mw = wrl[(KJF, "WRITE_SIZE")][1] * 1024
            txt += ["", "The masks: the forward kernel writes %.0f B per jacobian point (algorithmic: 512 B of relu bits + 4 B of sdf), the backward kernel reads them back and writes the "
                        "68-float rows (272 B per point): %.2f GB/s of fabric traffic in the pair -- %.1f %% of the HBM peak; MFMA-bound like every decoder kernel here." % (
                            mw / max(jac_pts, 1.0), (fel[(KJF, "FETCH_SIZE")][1] + fel[(KJB, "FETCH_SIZE")][1]) * 2048 / ((dfl[KJF][1] + dfl[KJB][1]) * 1e-3) / 1e9,
                            100 * (fel[(KJF, "FETCH_SIZE")][1] + fel[(KJB, "FETCH_SIZE")][1]) * 2048 / ((dfl[KJF][1] + dfl[KJB][1]) * 1e-3) / 8e12), ""]
    open(os.path.join(DST, TAG + "_pmc.md"), "w").write("\n".join(txt))
    traffic = {
        "fwd_fetch_bytes_per_point": round(k1_fetch / k1_pts, 1),
        "lp_fetch_bytes_per_point": round(k0_fetch / k0_pts, 1),
        "source": "profiles/" + TAG + "_pmc.md",
        "note": ("rocprofv3 --pmc FETCH_SIZE pass at the bench configuration, 64 objects per GPU (profiles/" + TAG + "_pmc.md): %.2f KB of L2-fabric reads per point "
                 "the fp32 forward kernel decodes (x2 gfx950 correction applied) = Infinity-Cache-served re-reads of the 6.8 MB fp32 weight stream, %.0f GB/s = "
                 "%.1f %% of the HBM peak; algorithmic 20 B/point.  Prepass kernel: %.0f B/point (its 3.6 MB f16 stream fits the 4 MiB L2)"
                 % (k1_fetch / k1_pts / 1e3, k1_fetch / (df[K1][1] * 1e-3) / 1e9, 100 * k1_fetch / (df[K1][1] * 1e-3) / 8e12, k0_fetch / k0_pts)),
    }
    json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)

    # ---- latency -----------------------------------------------------------------------------------------------------------------
    ls = stats_rows("latency_kernel_stats.md")
    runs = [v["calls"] for k, v in ls.items() if "k_solve" in k][0] // 10
    per_run = sum(v["total_ms"] for v in ls.values()) / runs
    tail = [l for l in read("latency_run.txt").splitlines() if not l.startswith("W2") and not l.startswith("E2") and l.strip()][-4:]
    n_it = [v for k, v in ls.items() if "k_solve" in k][0]["calls"]

    def per_it(sub):      # us per Gauss-Newton iteration spent in kernels whose name contains `sub`
        return 1e3 * sum(v["total_ms"] for k, v in ls.items() if sub in k and "ILb1EEEvNS_6LpArgs" not in k) / n_it

    book = sum(v["total_ms"] for k, v in ls.items() if not any(x in k for x in ("mlp_", "k_solve", "__amd", "k_init_state", "k_finalize", "k_code_bias")))
    ltxt = ["# Round %s -- single-detection latency path, rocprofv3 kernel stats" % ROUND, "",
            "`cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -- python tools/gpu_small_loop.py <M> <Bg> <reps>`: a resident batch of ONE object",
            "re-run `reps`+1 times (10 joint Gauss-Newton iterations each); tables by `tools/rocpd_stats.py`, this file by `tools/make_profiles.py`.",
            "Automatic kernel choice (no setters).  The object of this loop is seed 1 at detection size: some of its iterations keep more than 128",
            "jacobian tiles, which then take the latency form (`mlp_split_kernel<2>`) instead of the cluster form -- both kernels are launched every",
            "iteration and one of them returns at once (its ~4.5 us minimum in the table).", "",
            "## Real-KITTI-size detection: 250 surface points + 200 background rays (450 rays x 50 samples), %d runs -- %.2f ms of kernels per run under the profiler"
            % (runs, per_run),
            "## (%.2f ms p50 in `bench.py`, un-profiled, on the bench's own detection, whose lists all fit the cluster form; round 3: 5.43 ms, round 1: 11.06 ms)" % b["latency_kitti_size_ms_p50"], "",
            table_only("latency_kernel_stats.md"), "", "```"] + tail + ["```", "",
            "Per iteration (%d iterations), us: prepass `mlp_lp_kernel` %.0f; jacobian launch -- cluster form %.0f + latency form %.0f (one of the two does the work);"
            % (n_it, per_it("mlp_lp_kernel"), per_it("mlp_cluster_kernel"), per_it("mlp_split_kernel")),
            "`k_solve` %.0f (fp64 elimination with rows in lanes; round 3: 74); `k_gram` %.0f + `k_gram_reduce` %.0f; `k_render_scan` %.0f; `k_render_tail_wave` %.0f; `k_front_wave` %.0f; `k_band_wave` %.0f;"
            % (per_it("k_solve"), per_it("k_gramE"), per_it("k_gram_reduce"), per_it("k_render_scan"), per_it("k_render_tail_wave"), per_it("k_front_wave"), per_it("k_band_wave")),
            "`k_build_tiles` %.0f (a one-object batch runs without tile lists).  Everything that is not a decoder launch or the solve: **%.0f us per iteration = %.2f ms per call** (round 3: 126 us / 1.26 ms)."
            % (per_it("k_build_tiles"), 1e3 * book / n_it, book / runs), "",
            "## cfg2-size object: 2000 surface points + 500 background rays (2500 rays) -- %.2f ms p50 in `bench.py` (round 3: 17.30)" % b["latency_ms_p50"], "",
            "(the forward launch exports relu masks -- `mlp_kernel<1>` + its tail round as `mlp_split_kernel<1>` -- and the kept render rows run backward-only",
            "inside the jacobian launch `mlp_split_kernel<2>`: mixed mask reuse)", "",
            table_only("latency_cfg2_kernel_stats.md"), ""]
    open(os.path.join(DST, TAG + "_latency_kernel_stats.md"), "w").write("\n".join(ltxt))
    if os.path.exists(os.path.join(SRC, "latency_ab.txt")):
        ab = ["# Round %s -- per-detection latency, A/B of every switch on ONE box" % ROUND, "",
              "`python tools/gpu_latency_ab.py 15`: host wall clock p50 / min over 15 runs around `run + results` of a resident one-object batch (and the one-shot",
              "entry point, host buffers in).  Every variant returns the automatic path's bits (last column).", "", "```", read("latency_ab.txt").rstrip(), "```", "",
              "one-shot = `dsp_reconstruct_batch` (what `Optimizer.reconstruct_object` calls): build the batch out of the handle's pools, upload through pinned staging,",
              "run, ONE read-back, drop.", ""]
        open(os.path.join(DST, TAG + "_latency_ab.md"), "w").write("\n".join(ab))
    print("wrote profiles/%s_{bench_lines,kernel_stats,pmc,latency_kernel_stats}.md and pmc_traffic.json" % TAG)
    print("K1 busy %.1f%% clk %.2f; K2 %.1f%%; K2r %.1f%%; K0 busy %.1f%% clk %.2f" % (100 * busy(K1), clk(K1), 100 * busy(K2), 100 * busy(K2R), 100 * busy(K0), clk(K0)))
    print("K1 fetch B/pt %.0f  K0 fetch B/pt %.0f write B/pt %.0f" % (k1_fetch / k1_pts, k0_fetch / k0_pts, wr[(K0, "WRITE_SIZE")][1] * 1024 / k0_pts))


if __name__ == "__main__":
    main()
