#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite database, or *_kernel_stats.csv) as a small
markdown/CSV table for profiles/.  Usage: python tools/rocprof_summary.py <results.db|dir> [out.md]"""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    return [(r[0], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6) for r in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6,
                        float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
    return sorted(out, key=lambda r: -r[2])


def main():
    src = sys.argv[1]
    if os.path.isdir(src):
        dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
        csvs = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
        rows = from_db(dbs[0]) if dbs else from_csv(csvs[0])
    else:
        rows = from_db(src) if src.endswith(".db") else from_csv(src)
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx in rows[:40]:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {calls} | {tot:.3f} | {avg:.4f} | {mn:.4f} | {mx:.4f} | {100 * tot / total:.1f} |")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
