# L2 <-> memory write-path counters of C5's scatter kernel, with and without the hot window (two separate --pmc passes per mode,
# kernel trace only, as the guide prescribes).  usage: bash tools/gpu/r4_c5_pmc.sh <tag>
TAG=${1:-r4e}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in hot nohot; do
  if [ $mode = nohot ]; then export LIBGDF_AMD_LAB=1 GDF_GBP_NO_HOT=1; else unset LIBGDF_AMD_LAB GDF_GBP_NO_HOT; fi
  for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_TAG_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_DRAM_sum"; do
    n=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${mode}_$n -o x -- python $R/tools/bench_c5.py --reps 1 --no-checks > $O/pmc_${mode}_$n.log 2>&1
  done
done
cd $R
python - <<'PY' $O > $O/c5_scatter_l2_counters.txt
import csv, glob, sys, collections
O = sys.argv[1]
for mode in ("hot", "nohot"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter()
    for f in glob.glob(f"{O}/pmc_{mode}_*/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "gbp_scatter" not in k and "gbp_count" not in k and "gb_part_aggregate" not in k: continue
            k = k.split("<")[0].replace("gdf_amd::", "").replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (r["Dispatch_Id"], f) not in seen: seen.add((r["Dispatch_Id"], f)); launches[(k, r["Counter_Name"])] += 1
    print(f"== mode {mode} (per launch; bench_c5 --reps 1 = warm-up call + 1 timed call)")
    for k in sorted(acc):
        print(" ", k, {c: round(v / max(1, launches[(k, c)])) for c, v in sorted(acc[k].items())})
PY
cat $O/c5_scatter_l2_counters.txt
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
