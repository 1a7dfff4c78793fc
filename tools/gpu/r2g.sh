O=$GRAFT_REPO_ROOT/gpurun_out/r2g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o j -- python $GRAFT_REPO_ROOT/tools/bench_shapes.py --only c3_wide_keys --reps 1 > $O/log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.expandvars("$GRAFT_REPO_ROOT/gpurun_out/r2g/t/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
for r in rows:
    n = r["Kernel_Name"]
    if "jk_" in n:
        print(n[:70], "grid", r.get("Grid_Size"), r.get("Grid_Size_X"), "wg", r.get("Workgroup_Size"), r.get("Workgroup_Size_X"), "lds", r.get("LDS_Block_Size"), "scratch", r.get("Scratch_Size"), "vgpr", r.get("VGPR_Count"), "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
rm -rf $O/t
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hash_partition.py -x -q -m gpu -k "prefixsum" > $O/pytest_scan.txt 2>&1; tail -3 $O/pytest_scan.txt
timeout 300 python tools/bench_ops.py --ops scan 2>>$O/err.txt | cut -c1-400
