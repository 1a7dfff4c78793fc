#!/usr/bin/env python
"""Build profiles/*_pmc_hbm.json (read by bench.py's roofline.traffic) from two rocprofv3 PMC passes.
Usage: python tools/pmc_hbm_json.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> ["command text"]

FETCH_SIZE / WRITE_SIZE are reported in KB.  On gfx950 FETCH_SIZE reports half of a coalesced streaming read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section; calibrated in round 1 on jk_hist, a pure read of a known
8.8e9 B), so corrected bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024."""
import csv
import json
import re
import sys
from collections import defaultdict


def collect(path, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            m = re.search(r"gdf_amd::(\w+)(<\w+)?", name)
            short = m.group(1) if m else name.split("(")[0][-48:]
            if short == "jk_probe":       # jk_probe<WRITE, NARROW>: the label bench.py / GDF_LAUNCH use
                short = "jk_probe_write" if (m.group(2) or "") == "<true" else "jk_probe_count"
            if short == "jk_probe_fast":  # the plain-join write pass is launched under the same label
                short = "jk_probe_write"
            tot[short] += float(r["Counter_Value"])
            disp[short].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}


def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, _ = collect(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for k in fetch:
        if not k.startswith(("jk_", "gj_", "scan_", "gb_", "gbp_", "rs_", "sg_", "hp_", "fj_", "stable_", "part_")):
            continue
        kernels[k] = {"fetch_kb_reported": fetch[k], "write_kb_reported": write.get(k, 0.0), "launches": nf[k],
                      "hbm_bytes_per_join_corrected": 2.0 * fetch[k] * 1024.0 + write.get(k, 0.0) * 1024.0}
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import library_build_id
    out = {"build_id": library_build_id(),      # the kernel sources these counters belong to; bench.py refuses any other
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only), "
                     + (sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 1 --warmup 0 --cpu-sample 0 (one C3 join)"),
           "units": "FETCH_SIZE/WRITE_SIZE in KB as reported; corrected bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
                    "(gfx950: FETCH_SIZE counts half of a coalesced streaming read; calibrated on jk_hist)",
           "kernels": kernels}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_join_corrected"]):
        print(f"{k:28s} launches {v['launches']:3d}  HBM bytes {v['hbm_bytes_per_join_corrected'] / 1e9:8.3f} GB  per launch {v['hbm_bytes_per_join_corrected'] / max(v['launches'], 1) / 1e9:8.3f} GB")


if __name__ == "__main__":
    main()
