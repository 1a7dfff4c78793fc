#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel: sum of each counter over dispatches.
Usage: python tools/pmc_summary.py <counter_collection.csv> [name-filter]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(set)
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        short = name.split("(")[0][-40:]
        acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[short].add(r["Dispatch_Id"])
for k, v in acc.items():
    print(k, "dispatches", len(ndisp[k]))
    for c, val in sorted(v.items()):
        print(f"   {c:28s} {val:16.0f}   per-dispatch {val / len(ndisp[k]):14.0f}")
