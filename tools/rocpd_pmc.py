#!/usr/bin/env python3
"""Per-kernel PMC sums from a rocprofv3 (ROCm 7.2 rocpd sqlite) counter-collection run:
   python tools/rocpd_pmc.py <results.db> [kernel-name-substring] [--between A B]
--between A B: only the dispatches between the A-th and the B-th dispatch of bench.py's marker kernel (k_debug_lie, 1-based; bench.py opens
every leg of its run with one) -- e.g. `--between 1 3` = the headline leg's warm-up + timed steps, without the one-object latency probes
whose sub-millisecond launches of the same kernels would otherwise be averaged in."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    window = ""
    if "--between" in sys.argv:
        i = sys.argv.index("--between")
        lo, hi = int(sys.argv[i + 1]), int(sys.argv[i + 2])
        marks = [r[0] for r in cur.execute("select d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                                           "where s.kernel_name like '%k_debug_lie%' order by d.start")]
        if len(marks) < hi:
            raise SystemExit("only %d marker dispatches in the trace" % len(marks))
        window = " and d.start >= %d and d.start < %d" % (marks[lo - 1], marks[hi - 1])
        print("(dispatches between marker %d and marker %d of %d)" % (lo, hi, len(marks)))
    # pmc_event.event_id -> rocpd_event.id ; kernel dispatch rows carry event_id too
    cols = [c[1] for c in cur.execute("pragma table_info('rocpd_kernel_dispatch')")]
    key = "event_id" if "event_id" in cols else "id"
    q = ("select s.kernel_name, p.name, count(distinct d.id), sum(e.value), sum(d.end - d.start) / count(distinct p.name || e.id) * 1.0 "
         "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on d.%s = e.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "where s.kernel_name like ?%s group by s.kernel_name, p.name order by s.kernel_name, p.name" % (key, window))
    rows = cur.execute(q, ("%" + sub + "%",)).fetchall()
    print("| kernel | counter | dispatches | sum | per dispatch |")
    print("|---|---|---|---|---|")
    for name, pmc, n, total, _ in rows:
        short = name if len(name) < 60 else name[:57] + "..."
        print("| %s | %s | %d | %.6g | %.6g |" % (short, pmc, n, total, total / max(n, 1)))
    # durations per kernel for rate computations
    rows = cur.execute("select s.kernel_name, count(*), sum(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id where s.kernel_name like ?" + window + " group by s.kernel_name", ("%" + sub + "%",)).fetchall()
    for name, n, ns in rows:
        print("duration: %s  dispatches %d  total %.3f ms" % (name[:60], n, ns / 1e6))


if __name__ == "__main__":
    main()
