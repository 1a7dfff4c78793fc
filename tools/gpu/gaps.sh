# idle gaps between consecutive kernels of one C3 join (rocprofv3 kernel trace timestamps)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gaps; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o j -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --pandas-sample 0 > $O/log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.expandvars("$GRAFT_REPO_ROOT/gpurun_out/gaps/t/**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last join = from the last jk_hist to the end
last = max(i for i, n in enumerate(names) if "jk_hist" in n)
prev_end = None
tot_gap = 0
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    tot_gap += gap
    print(f"{r['Kernel_Name'][:60]:60s} dur {(e - s) / 1e3:9.1f} us   gap before {gap:8.1f} us")
    prev_end = e
print("total gap us", tot_gap, "span ms", (int(rows[-1]["End_Timestamp"]) - int(rows[last]["Start_Timestamp"])) / 1e6)
PY
rm -rf $O/t
