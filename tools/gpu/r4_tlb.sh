# is jk_scatter1's slow mode a matter of address translation?  UTCL1 / UTCL2 counters of four processes (each lands in one of the two modes);
# kernel durations come from the same runs' kernel trace
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4tlb}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
true
true
true
B="python $R/bench.py --cpu-sample 0 --pandas-sample 0 --extra 0 --steps 2 --warmup 1"
for i in 1 2 3 4 5; do
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum --kernel-trace --output-format csv -d $O/p$i -o j -- $B > $O/p$i.log 2>&1
  python $R/tools/pmc_summary.py $(find $O/p$i -name "*counter_collection.csv" | head -1) 2>/dev/null | grep -A4 -E "jk_scatter1|jk_scatter2|jk_probe_fast" | head -40 > $O/p$i.txt
  python - <<PY >> $O/p$i.txt
import csv, glob
f = glob.glob("$O/p$i/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows:
    n = r.get("Kernel_Name", "")
    if "jk_scatter1" in n or "jk_scatter2" in n or "jk_probe_fast" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d > 1.0: print("duration_ms", n[:60], round(d, 3))
PY
  cat $O/p$i.txt
done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
