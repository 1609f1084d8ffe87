#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into the per-kernel stats table that
`rocprofv3 --stats` prints in CSV mode:  python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds in rows:
        short = name if len(name) < 70 else name[:67] + "..."
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s |" % (short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, sg, lds))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
