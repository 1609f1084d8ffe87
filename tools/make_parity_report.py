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
    # keep the latest record per (kind, case)
    last = {}
    for r in recs:
        last[(r["kind"], r.get("case", ""), r.get("dtype", ""))] = r
    print("# Parity report -- measured on MI355X by `pytest -m gpu` (tests/conftest.py:parity_log)\n")
    print("Every number is |device - reference| (or oracle) as the test measured it, next to the bound it was held to.\n")
    for r in [r for (k, _, _d), r in last.items() if k == "build_info"]:
        print("Library that ran: `%s` -- %s; HIP runtime %s, driver %s.\n" % (r["lib"], r["info"], r["hip_runtime"], r["hip_driver"]))
    ars = [r for (k, _, _d), r in last.items() if k == "at_reference_states"]
    if ars:
        print("## Device linearised at the REFERENCE'S OWN recorded states (pose, code and depth samples injected bit for bit), compared with the reference's recorded H / b / dx / V / K directly\n")
        print("Every iteration of every recorded run (tests/test_gpu_forensics.py::test_linearisation_at_reference_states).  strict = V and K identical to the reference's, H within 3e-5 and b within 1.2e-4 of the reference's recorded values (3x what was measured).\n")
        print("| golden | iterations | strict | iterations with named flips | max rel dH | max rel db | max rel d(dx) | K per iteration |")
        print("|---|---|---|---|---|---|---|---|")
        for r in sorted(ars, key=lambda r: r["case"]):
            print("| %s | %d | %d | %d | %.2e | %.2e | %.2e | %s |" % (r["case"], r["n"], r["strict"], sum(1 for f in r["flips"] if f), max(r["rel_H"]), max(r["rel_b"]),
                                                                   max(r["rel_dx"]), r["K"]))
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
        print("Bound per quantity: max(1e-4, the reference's own spread when every input element moves to an adjacent float32: 9 draws, golden ulp*_ fields).\n")
        print("| golden | rot abs (reference spread) | scale rel (spread) | trans rel (spread) | code abs (spread) | loss rel | max abs dT (spread) |")
        print("|---|---|---|---|---|---|---|")
        for r in sorted(e2e, key=lambda r: r["case"]):
            print("| %s | %.2e (%.2e) | %.2e (%.2e) | %.2e (%.2e) | %.2e (%.2e) | %.2e | %.2e (%.2e) |" % (
                r["case"], r["rot"], r["rot_sens"], r["scale"], r["scale_sens"], r["trans"], r["trans_sens"], r["code"], r["code_sens"],
                r["loss"], r["t_abs"], r["t_abs_sens"]))
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
