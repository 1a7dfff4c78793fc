#!/usr/bin/env python
"""Timeline of the LAST n kernels of a rocprofv3 --kernel-trace csv: start offset, duration and the idle gap in front of each
(us).  Usage: python tools/kernel_timeline.py <dir with *_kernel_trace.csv> [n]"""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
