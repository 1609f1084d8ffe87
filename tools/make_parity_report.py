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
