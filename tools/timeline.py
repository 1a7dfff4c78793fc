"""timeline.py <dir with *_kernel_trace.csv> [marker-kernel-substring] -- the launches of the LAST call of an operator as a timeline:
offset from the call's first launch, duration, and the idle gap in front of every launch (host round trips show up here).
A call is delimited by its first kernel (default: jk_sample_skew, the first launch of gdf_inner_join)."""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "jk_sample_skew"
    path = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if not starts:
        print("no launch of", marker)
        return
    calls = []
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        calls.append(rows[a:b])
    # the last complete call: cut at the first launch that is not the library's (torch kernels of the next step)
    call = calls[-2] if len(calls) > 1 else calls[-1]
    lib = [r for r in call if "gdf_amd" in r[2] or "rocclr" in r[2]]
    t0 = lib[0][0]
    prev_end = t0
    busy = 0.0
    gaps = 0.0
    print("| offset us | duration us | gap in front us | kernel |\n|---|---|---|---|")
    for s, e, name in lib:
        gap = s - prev_end
        short = name.replace("void gdf_amd::", "").replace("gdf_amd::", "")[:70]
        print("| %.1f | %.1f | %.1f | `%s` |" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short))
        busy += (e - s)
        gaps += max(gap, 0.0)
        prev_end = max(prev_end, e)
    print("\nspan %.3f ms, busy %.3f ms, idle %.3f ms over %d launches" % ((prev_end - t0) / 1e6, busy / 1e6, gaps / 1e6, len(lib)))


if __name__ == "__main__":
    main()
