#!/usr/bin/env python3
"""gpurun_out/parity/*.jsonl (one file per pytest session, written by the -m gpu parity tests: tests/conftest.py:parity_log) -> a
markdown report with the latest record per case:
   python tools/make_parity_report.py [gpurun_out/parity | file.jsonl] > profiles/parity_rNN.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity")
    files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith(".jsonl")) if os.path.isdir(path) else [path]
    legacy = os.path.join(ROOT, "gpurun_out", "parity.jsonl")      # single-file form of earlier runs
    if os.path.isdir(path) and os.path.exists(legacy):
        files = [legacy] + files
    recs = sorted((json.loads(l) for fn in files for l in open(fn) if l.strip()), key=lambda r: r.get("_t", 0.0))
    for r in recs:
        r.setdefault("kind", "lie" if str(r.get("case", "")).startswith("lie_") else "other")
    # keep the latest record per (kind, case)
    last = {}
    for r in recs:
        last[(r["kind"], r.get("case", ""), r.get("dtype", ""))] = r
    print("# Parity report -- measured on MI355X by `pytest -m gpu` (tests/conftest.py:parity_log)\n")
    print("Every number is |device - reference| (or oracle) as the test measured it, next to the bound it was held to.\n")
    for r in [r for (k, _, _d), r in last.items() if k == "build_info"]:
        print("Library that ran: `%s` -- %s; HIP runtime %s, driver %s.\n" % (r["lib"], r["info"], r["hip_runtime"], r["hip_driver"]))
    lie = [r for (k, _, _d), r in last.items() if k == "lie"]
    if lie:
        print("## Row a11 on the device: the Lie maps, the rotation prior and the Sim(3) update evaluated by `dsp_debug_lie` (the device functions `k_solve` calls) against vectors recorded from the reference\n")
        print("tests/test_gpu_lie.py; `golden_terms.npz` (tools/make_golden.py), `golden_lie.npz` (tools/make_golden_lie.py).  Errors in float32 ulp OF THE MATRIX'S LARGEST ENTRY (bound: 2; the state update: 4).\n")
        print("| case | vectors | worst error (ulp) | branches executed on the device |")
        print("|---|---|---|---|")
        for r in sorted(lie, key=lambda r: r["case"]):
            if "exp_sim3_ulp" in r:
                print("| %s | %d | exp_sim3 %.2f, exp_se3 %.2f | %s |" % (r["case"], r["n"], r["exp_sim3_ulp"], r["exp_se3_ulp"], ", ".join(r["branches"])))
            elif "res_ulp" in r:
                print("| %s | %d | residual %.2f, J %.2f | zero branch (res < 1e-7) on %d poses; %d pose(s) within round-off of the threshold |" % (
                    r["case"], r["n"], r["res_ulp"], r["j_ulp"], r["zero_branch"], r["threshold_edge"]))
            else:
                print("| %s | %d | %.2f | exp_sim3(dx) @ t_obj_cam |" % (r["case"], r["n"], r["ulp"]))
        print()
    cx = [r for (k, _, _d), r in last.items() if k in ("complex_calibration", "complex_bench")]
    if cx:
        print("## The decoder of realistic complexity (`decoder_complex.npz`: non-convex multi-part shape family, codes on all 64 dimensions)\n")
        for r in cx:
            if r["kind"] == "complex_calibration":
                print("f16 prepass calibration at `dsp_create` (32 768 unit-ball points x 2 codes per magnitude):\n")
                print("| code magnitude (uniform on every entry) | " + " | ".join("%.2f" % m for m in r["mags"]) + " |")
                print("|---|" + "---|" * len(r["mags"]))
                print("| largest abs(sdf_f16 - sdf_fp32) | " + " | ".join("%.2e" % e for e in r["max_err"]) + " |")
                print("| margin delta | " + " | ".join("%.2e" % e for e in r["delta"]) + " |\n")
            else:
                print("64 cfg2-size objects of the complex family (tests/test_gpu_complex.py::test_complex_prepass_is_exact_and_pays): **%.1f objects/s with the prepass, %.1f without** "
                      "(bit-identical results, %d of %d objects good); %.1f %% of the in-sphere samples reach the fp32 kernel; audit: %d misclassified of %.3g samples, largest prepass error %.2e; "
                      "guard: largest error seen %.2e, no trip; fp32 forward kernel %.3f of peak (%.3f with the prepass off), prepass kernel %.0f TFLOP/s.\n" % (
                          r["objects_per_s_prepass_on"], r["objects_per_s_prepass_off"], r["objects_good"], r["objects"], 100 * r["fwd_points_evaluated_over_insphere"], 0,
                          r["audited"], r["audit_max_err"], r["guard_max_err"], r["fwd_fp32_frac"], r["prepass_off_fwd_fp32_frac"], r["prepass_tflops"]))
    co = [r for (k, _, _d), r in last.items() if k == "cotenant"]
    if co:
        print("## The per-detection path beside a co-tenant on the same GPU (tests/test_gpu_cotenant.py; profiles/r05_latency_cotenant.md)\n")
        print("| co-tenant | solo p50 ms | shared p50 / p99 / max ms | runs | fell back to the latency form on the device | bits equal to the solo run |")
        print("|---|---|---|---|---|---|")
        for r in sorted(co, key=lambda r: r["case"]):
            print("| %s | %.2f | %.2f / %.2f / %.2f | %d | %d | yes (asserted) |" % (r["case"], r["solo_p50_ms"], r["shared_p50_ms"], r["shared_p99_ms"], r["shared_max_ms"], r["runs"], r["fallback_runs"]))
        print()
    ars = [r for (k, _, _d), r in last.items() if k == "at_reference_states"]
    if ars:
        print("## Device linearised at the REFERENCE'S OWN recorded states (pose, code and depth samples injected bit for bit), compared with the reference's recorded H / b / dx / V / K directly\n")
        print("Every iteration of every recorded run (tests/test_gpu_forensics.py::test_linearisation_at_reference_states).  strict = V and K identical to the reference's, H within 3e-5 and b within 1.2e-4 of the reference's recorded values (3x what was measured).\n")
        print("| golden | iterations | strict | iterations with named flips | max rel dH | max rel db | max rel d(dx) | max rel d(loss) | K per iteration |")
        print("|---|---|---|---|---|---|---|---|---|")
        for r in sorted(ars, key=lambda r: r["case"]):
            print("| %s | %d | %d | %d | %.2e | %.2e | %.2e | %s | %s |" % (r["case"], r["n"], r["strict"], sum(1 for f in r["flips"] if f), max(r["rel_H"]), max(r["rel_b"]),
                                                                        max(r["rel_dx"]), "%.2e" % max(r["rel_loss"]) if "rel_loss" in r else "-", r["K"]))
        n_loss = sum(len(r["rel_loss"]) for r in ars if "rel_loss" in r)
        if n_loss:
            print("\n`loss` = the result's fourth field after ONE iteration from the recorded state = the reference's loss AT that state (reconstruct/optimizer.py:155; the field "
                  "src/LocalMapping_util.cc:405-406 branches on); `it_loss` from the reference's own functions at the recorded state (tools/make_golden_it_loss.py; its last entry is "
                  "bit-identical to the loss the recorded run returned).  **%d loss comparisons, bound 1e-4 (asserted).**" % n_loss)
        print("\n(rel d(dx) is dominated by the three rotation-prior entries of b: k4 = 1e7 times a residual that is a difference of two numbers ~1, i.e. float32 ulp noise in the reference itself; the test bounds dx through |H^-1| with that term.)\n")
    chf = [r for (k, _, _d), r in last.items() if k == "chained_forensic"]
    if chf:
        print("## Chained run beside the reference's recorded trajectory: every step's difference decomposed\n")
        print("local = device step vs oracle step FROM THE DEVICE'S OWN STATE (asserted: identical sets or named flips, dx to 1e-4 through |H^-1|); propagated = oracle step from the device's state vs the "
              "reference's recorded step = the MAP's response to the state difference that came in.  Per-case tables with the named samples: profiles/r03_forensics_*.md.\n")
        print("| golden | first iteration whose sets differ from the recorded ones | max local rel d(dx) | max propagated rel d(dx) | incoming state diff at that iteration | samples named (explained) | final rot / scale / trans / code | reference's own 1-ulp spread |")
        print("|---|---|---|---|---|---|---|---|")
        for r in sorted(chf, key=lambda r: r["case"]):
            f = r["first_flip_iteration"]
            m, sp = r["final"], r["reference_spread"]
            print("| %s | %s | %.2e | %.2e | %s | %d (%d) | %.1e / %.1e / %.1e / %.1e | %.1e / %.1e / %.1e / %.1e |" % (
                r["case"], "none" if f is None else f, max(r["local_rel_dx"]), max(r["propagated_rel_dx"]), "-" if f is None else "%.1e" % r["incoming_state_diff"][f],
                r["flips_named"], r["flips_explained"], m["rot"], m["scale"], m["trans"], m["code"], sp["rot"], sp["scale"], sp["trans"], sp["code"]))
        print()
    bar = [r for (k, _, _d), r in last.items() if k == "bench_at_reference_states"]
    bar = sorted(bar, key=lambda r: r.get("_t", 0.0))[-1:]           # the latest run (the case label carries the number of traced objects)
    if bar:
        print("## The HEADLINE workload (64 x cfg2 bench batch) at the reference's own recorded states, inside the resident 64-object batch\n")
        print("tests/test_gpu_bench_objects.py::test_batch64_at_the_references_recorded_states: `tests/golden/golden_bench_cfg2x64.npz` (made by `tools/make_golden_bench.py` "
              "from the unmodified reference) holds all ten iterations of sixteen of bench.py's 64 objects; each traced object's recorded pose, code and depth samples "
              "are injected into ITS slot of the 64-object batch (the other 56 run on), one Gauss-Newton iteration is taken and V, K, H, b are compared with the "
              "reference's recorded values.  strict = V and K identical, H within 3e-5, b within 1.2e-4; the rotation-prior block H[3:6,3:6] is held to "
              "`k4 (j_i + j_j) 1e-6` on top (the reference builds it in float32 from a residual that is a difference of numbers ~1 and multiplies by k4 = 1e7: "
              "the oracle differs from the recording by the same amount).\n")
        for r in bar:
            print("%s: **%d iterations, %d strict, %d with named flips**; max rel dH %.2e, max rel db %.2e%s.\n" % (r["case"], r["n"], r["strict"], r["with_named_flips"],
                                                                                                                  r["max_rel_H"], r["max_rel_b"],
                                                                                                                  "; **%d loss comparisons, max rel d(loss) %.2e** (bound 1e-4, asserted)" % (
                                                                                                                      r["n_loss_comparisons"], r["max_rel_loss"]) if "max_rel_loss" in r else ""))
            print("| bench object | max rel dH | max rel db | max rel d(loss) | named flips | K per iteration (identical to the reference's) |")
            print("|---|---|---|---|---|---|")
            for o in sorted(r["per_object"], key=int):
                q = r["per_object"][o]
                print("| %s | %.2e | %.2e | %s | %d | %s |" % (o, q["max_rel_H"], q["max_rel_b"], "%.2e" % q["max_rel_loss"] if "max_rel_loss" in q else "-", q["named"], q["K"]))
            if r.get("beyond_tight_bounds"):
                print("\nIterations beyond 3e-5 / 1.2e-4 and what the oracle's own response to a 1-ulp state jitter is there (the bound they were held to instead):\n")
                for x in r["beyond_tight_bounds"]:
                    print("- object %d iteration %d: rel dH %.2e, rel db %.2e; oracle under jitter: rel dH %.2e, rel db %.2e%s" % (
                        x["object"], x["iteration"], x["rel_H"], x["rel_b"], x["oracle_jitter_rel_H"], x["oracle_jitter_rel_b"],
                        " (the jitter flips a sample)" if x.get("oracle_jitter_flips_a_sample") else ""))
            print()
    bch = [r for (k, _, _d), r in last.items() if k == "bench_chained"]
    if bch:
        print("## The headline workload chained: all 64 bench objects, ten iterations, against the reference's recorded results\n")
        for r in bch:
            print("tests/test_gpu_bench_objects.py::test_batch64_first_iteration_and_chained_result_vs_reference: iteration 0 -- %d of %d objects with sample sets IDENTICAL to the reference's, the rest within "
                  "dV <= %d, dK <= %d (asserted: <= 2 and >= B - 8 exact); after ten iterations the traced objects are held to 1.5x the reference's OWN spread "
                  "under a 1-ulp jitter of its inputs (recorded in the golden), the others to 3x the worst traced spread.\n" % (
                      r["objects_identical_sets_iteration0"], r["n_objects"], r["max_dV_iteration0"], r["max_dK_iteration0"]))
            print("| traced bench object | device vs reference: rot / scale / trans / code | reference's own 1-ulp spread: rot / scale / trans / code |")
            print("|---|---|---|")
            for o in sorted(r["traced"], key=int):
                m, sp = r["traced"][o]["measured"], r["traced"][o]["reference_spread"]
                print("| %s | %.1e / %.1e / %.1e / %.1e | %.1e / %.1e / %.1e / %.1e |" % (o, m["rot"], m["scale"], m["trans"], m["code"], sp["rot"], sp["scale"], sp["trans"], sp["code"]))
            rest = {k: v for k, v in r.items() if k not in ("kind", "case", "traced", "_t", "objects_identical_sets_iteration0", "n_objects", "max_dV_iteration0", "max_dK_iteration0")}
            if rest:
                print("\nOther recorded figures: `%s`" % json.dumps(rest)[:900])
            print()
    mf = {k: r for (k, _, _d), r in last.items() if k in ("mono_flip_at_reference_states", "mono_flip_chained")}
    if mf:
        print("## The field the mono path branches on: both hypotheses of one detection (tests/test_mono_flip.py, `golden_mono_flip.npz`)\n")
        if "mono_flip_at_reference_states" in mf:
            r = mf["mono_flip_at_reference_states"]
            print("At the reference's recorded states of BOTH runs (%d linearisations): max rel d(loss) %.2e (detection's pose) / %.2e (yaw-flipped), V and K identical, bound 1e-4 (asserted).\n" % (
                r["n"], max(r["rel_loss_a"]), max(r["rel_loss_b"])))
        if "mono_flip_chained" in mf:
            r = mf["mono_flip_chained"]
            print("Chained through the mirror's `Optimizer.reconstruct_object` (five-argument form, as `LocalMapping_util.cc:391-406` calls it): device losses %.6f / %.6f, reference %.6f / %.6f "
                  "(rel %.1e / %.1e; the gap between the hypotheses is 0.28) -> C++ keeps the **%s** hypothesis on both: %s.\n" % (
                      r["device_loss"][0], r["device_loss"][1], r["reference_loss"][0], r["reference_loss"][1], r["rel"][0], r["rel"][1],
                      "flipped" if r["reference_loss"][0] > r["reference_loss"][1] else "unflipped", "same branch" if r["same_branch"] else "DIFFERENT BRANCH"))
    lpk = [r for (k, _, _d), r in last.items() if k == "lp_jacobian_kernel"]
    lps = [r for (k, _, _d), r in last.items() if k == "lp_compute_at_reference_states"]
    lpc = [r for (k, _, _d), r in last.items() if k == "lp_compute_chained"]
    lpv = [r for (k, _, _d), r in last.items() if k == "lp_compute_speed"]
    if lpk or lps:
        print("## The OPT-IN low-precision compute mode (`dsp_batch_set_compute`: 16-bit MFMA operands, fp32 accumulation) -- NOT the parity path; measured against it and the reference\n")
        if lpk:
            print("Kernel level (tests/test_gpu_lp_compute.py): `mlp_lpj_fwd_kernel` + `mlp_lpj_bwd_kernel` against the fp32 kernel's d sdf / d [code, xyz] on the same points; the sdf equals the prepass kernel's bit for bit (asserted).\n")
            print("| case | abs(dg) / max abs(g): median | 99 % | max (a relu unit at zero flipped by the 16-bit forward) | max abs(dsdf) |")
            print("|---|---|---|---|---|")
            for r in sorted(lpk, key=lambda r: r["case"]):
                print("| %s | %.2e | %.2e | %.2e | %.2e |" % (r["case"], r["grad_rel_median"], r["grad_rel_p99"], r["grad_rel_max"], r["sdf_abs_max"]))
            print()
        for r in sorted(lps, key=lambda r: r["case"]):
            w, m = r["worst"], r["median"]
            print("**%s**, at the reference's recorded states (%d linearisations): V identical; abs(dK) <= %d (%.2f %% of K); rel dH median %.2e / max %.2e; rel db %.2e / %.2e; "
                  "rel d(dx) %.2e / %.2e; rel d(loss) %.2e / %.2e.\n" % (r["case"], r["n"], w["dK"], 100 * r["worst_dK_over_K"], m["rel_H"], w["rel_H"], m["rel_b"], w["rel_b"],
                                                                          m["rel_dx"], w["rel_dx"], m["rel_loss"], w["rel_loss"]))
        for r in lpc:
            print("Chained ten-iteration runs, f16 mode (final result vs the reference's; for scale the fp32 path's distance and the reference's own spread under 1-ulp inputs / other thread counts):\n")
            print("| golden | f16 mode: rot / trans / code | fp32 path: rot / trans / code | reference's own spread | rel d(loss) |")
            print("|---|---|---|---|---|")
            for x in r["rows"]:
                a, f, sp = x["lp_vs_reference"], x["fp32_vs_reference"], x["reference_spread"]
                print("| %s | %.1e / %.1e / %.1e | %.1e / %.1e / %.1e | %.1e / %.1e / %.1e | %.1e |" % (x["case"], a["rot"], a["trans"], a["code"], f["rot"], f["trans"], f["code"],
                                                                                                    sp["rot"], sp["trans"], sp["code"], x["loss_rel_vs_reference"]))
            print()
        for r in lpv:
            print("%s: decoder time per run %.1f ms (fp32 path with the f16 classifier) -> %.1f ms (f16 compute mode); whole run %.1f -> %.1f ms.\n" % (
                r["case"], r["decoder_ms_fp32_path"], r["decoder_ms_lp"], r["run_ms_fp32_path"], r["run_ms_lp"]))
    sv = [r for (k, _, _d), r in last.items() if k == "solve_vs_lapack"]
    for r in sv:
        print("## `k_solve` against float64 LAPACK\n\n%s: max rel d(dx) %.2e (what rounding the traced H, b to float32 allows: %.2e).\n" % (r["case"], r["rel_dx_max"], r["rounding_bound_rel"]))
    r4 = [r for (k, _, _d), r in last.items() if k in ("solver_ab", "partial_guard_rerun", "one_shot", "reoptimise_map")]
    if r4:
        print("## Round-4 paths\n")
        for r in sorted(r4, key=lambda r: r["kind"]):
            if r["kind"] == "solver_ab":
                print("- `k_solve<0>` (fp64 LDL^T, b as row n) vs `k_solve<1>` (fp64 Gauss-Jordan, round 3), 8 mixed objects x 10 iterations: rel d(dx) first iteration %.1e, "
                      "all iterations %.1e (float32 results identical).  The factorisation itself: tests/test_solve_emulation.py (CPU, reference's recorded H / b)."
                      % (r["rel_dx_first_iteration"], r["rel_dx_all_iterations"]))
            elif r["kind"] == "partial_guard_rerun":
                print("- per-object guard re-run (%s): prepass errors %s, forced margin %.3g -> objects %s tripped, %d re-run with the prepass off (the others keep "
                      "their results, bit-identical to a batch of their own)." % (r["case"], ["%.2e" % e for e in r["errs"]], r["delta"], r["expected_objects"], r["rerun_objects"]))
            elif r["kind"] == "one_shot":
                print("- one-shot entry point (%s): resident batch %.3f ms p50, `dsp_reconstruct_batch` with host buffers %.3f ms p50 (pooled workspace, pinned staging, one read-back)."
                      % (r["case"], r["resident_ms_p50"], r["one_shot_ms_p50"]))
            elif r["kind"] == "reoptimise_map":
                print("- `tools/reoptimise_map.py` (%s): %d objects re-optimised in %.4f s; sharded over two handles == unsharded: %s."
                      % (r["case"], r["n_good"], r["seconds"], r["sharded_equals_unsharded"]))
        print()
    pg = [r for (k, _, _d), r in last.items() if k == "prepass_guard"]
    if pg:
        print("## Prepass WITHOUT the audit: always-on guard (every re-decoded sample compared with the prepass value it replaces)\n")
        print("| case | dtype | largest abs(sdf_lp - sdf_fp32) the guard saw | margin of a zero code | guard trips | re-run with prepass off | bit-identical to prepass off |")
        print("|---|---|---|---|---|---|---|")
        for r in sorted(pg, key=lambda r: (r["case"], r["dtype"])):
            print("| %s | %s | %.3g | %.3g | %d | %s | %s |" % (r["case"], r["dtype"], r["guard_max_err"], r["delta_zero_code"], r["trips"], r.get("rerun", False), r["identical"]))
        print()
    e2e = [r for (k, _, _d), r in last.items() if k == "end_to_end"]
    if e2e:
        print("## Chained GN run vs the reference's final result (golden files recorded from the unmodified reference)\n")
        print("Bound per quantity: max(1e-4, 1.5 x the reference's own spread: every input element moved to an adjacent float32 (golden ulp*_ fields: 9 draws, 33 on the "
              "chaotic full-size fixtures cfg2 and complex) and, with NO input change, torch.set_num_threads(1 / 4) instead of 8 (thr_*), whichever is larger).\n")
        print("| golden | rot abs (reference spread) | scale rel (spread) | trans rel (spread) | code abs (spread) | loss rel | max abs dT (spread) |")
        print("|---|---|---|---|---|---|---|")
        for r in sorted(e2e, key=lambda r: r["case"]):
            print("| %s | %.2e (%.2e) | %.2e (%.2e) | %.2e (%.2e) | %.2e (%.2e) | %.2e | %.2e (%.2e) |" % (
                r["case"], r["rot"], r["rot_sens"], r["scale"], r["scale_sens"], r["trans"], r["trans_sens"], r["code"], r["code_sens"],
                r.get("loss", float("nan")), r["t_abs"], r["t_abs_sens"]))
        print()
    its = [r for (k, _, _d), r in last.items() if k == "iterations"]
    if its:
        print("## Every GN iteration re-linearised by the oracle from the device's own state\n")
        print("strict = identical sample sets (membership checksums) AND the oracle's own jitter response < 1e-3: compared at 1e-4 + 4 x jitter response.\n")
        print("| case | iterations | identical sets | strict | threshold flips per iteration | max rel dH | max rel db | oracle jitter response (rel H) | K per iteration |")
        print("|---|---|---|---|---|---|---|---|---|")
        for r in sorted(its, key=lambda r: r["case"]):
            print("| %s | %d | %d | %d | %s | %.2e | %.2e | %.2e | %s |" % (
                r["case"], r["n"], r["same_sets"], r["strict"], r["flips"], max(r["rel_H"]), max(r["rel_b"]), max(r["oracle_jitter_rel_H"]), r["K"]))
        print()
    b64 = [r for (k, _, _d), r in last.items() if k == "batch64"]
    for r in b64:
        print("## %s\n" % r["case"])
        print("| object | iteration | identical sets | flips | rel dH | rel db | oracle jitter response | V | K |")
        print("|---|---|---|---|---|---|---|---|---|")
        for c in r["checks"]:
            print("| %d | %d | %s | %d | %.2e | %.2e | %.2e | %d | %d |" % (c["object"], c["iteration"], c["same_sets"], c["flips"], c["rel_H"], c["rel_b"],
                                                                           c["oracle_jitter_rel_H"], c["V"], c["K"]))
        print()
    pp = [r for (k, _, _d), r in last.items() if k == "prepass"]
    if pp:
        print("## Prepass (low-precision classification): audit against the fp32 decoder\n")
        print("| case | dtype | samples audited | max abs(sdf_lp - sdf_fp32) | margin delta | misclassified | fp32 forward points / in-sphere | bit-identical to prepass off |")
        print("|---|---|---|---|---|---|---|---|")
        for r in sorted(pp, key=lambda r: (r["case"], r["dtype"])):
            print("| %s | %s | %.3g | %.3g | %.3g | %d | %.3f | %s |" % (r["case"], r["dtype"], r["audited"], r["max_err"], r["delta"], r["misclassified"],
                                                                       r["fwd_over_insphere"], r["identical"]))
        print()


if __name__ == "__main__":
    main()
