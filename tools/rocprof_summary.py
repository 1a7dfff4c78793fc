#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite database, or *_kernel_stats.csv) as a small
markdown/CSV table for profiles/.  Usage: python tools/rocprof_summary.py <results.db|dir> [out.md]

Round 5: the first call of a shape runs the placement tournaments -- a handful of CALIBRATION launches of the regroup /
probe / scatter kernels over a quarter of their work.  From a database they are recognised (a launch shorter than 45 % of the same
kernel's longest one, of a kernel on the list below) and reported on a line of their own, so that the per-kernel averages are those of
the full launches."""
import csv
import glob
import os
import sqlite3
import sys


CALIBRATED = ("jk_scatter1", "jk_scatter2", "jk_probe_fast", "jk_probe_bp", "gbp_scatter_static")
DROPPED = []


def from_db(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    per = {}
    for name, dur in c.execute(f"select {name_col}, end-start from kernels"):
        per.setdefault(name, []).append(float(dur))
    return split_calibration(per)


def split_calibration(per):
    """per: {kernel name: [launch durations, ns]} -> table rows; the calibration launches of the placement tournaments (a launch shorter
    than 45 % of the same kernel's longest one, kernels of CALIBRATED only) are taken out of the averages and listed in DROPPED"""
    rows = []
    for name, durs in per.items():
        if any(k in name for k in CALIBRATED) and len(durs) > 1:
            longest = max(durs)
            short = [d for d in durs if d < 0.45 * longest]
            if short and len(short) < len(durs):
                DROPPED.append((name, len(short), sum(short) / 1e6))
                durs = [d for d in durs if d >= 0.45 * longest]
        rows.append((name, len(durs), sum(durs) / 1e6, sum(durs) / len(durs) / 1e6, min(durs) / 1e6, max(durs) / 1e6))
    return sorted(rows, key=lambda r: -r[2])


def from_trace_csv(path):
    """*_kernel_trace.csv of `rocprofv3 --kernel-trace --output-format csv`: one row per dispatch"""
    per = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            per.setdefault(r["Kernel_Name"], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return split_calibration(per)


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
        traces = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
        rows = from_db(dbs[0]) if dbs else (from_trace_csv(traces[0]) if traces else from_csv(csvs[0]))
    else:
        rows = from_db(src) if src.endswith(".db") else from_csv(src)
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx in rows[:40]:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {calls} | {tot:.3f} | {avg:.4f} | {mn:.4f} | {mx:.4f} | {100 * tot / total:.1f} |")
    for name, n, ms in DROPPED:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| (calibration launches of the placement tournament, not in the row above) `{short}` | {n} | {ms:.3f} | {ms / n:.4f} | | | |")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
