O=gpurun_out/r2u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_groupby.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python tools/bench_shapes.py --only c2_dense_keys,c2_sparse_keys 2>>$O/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"
echo "--- bench dist tail"
timeout 600 python bench.py --force-distributed --strategy fused --steps 3 --warmup 1 --probe-rows 1000000000 --build-rows 125000000 2>>$O/err.txt | tail -1 | cut -c1-200
# kernel timeline of one sparse C2 call
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/t -o j -- python $GRAFT_REPO_ROOT/tools/bench_shapes.py --only c2_sparse_keys --reps 2 > $GRAFT_REPO_ROOT/$O/log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.expandvars("$GRAFT_REPO_ROOT/gpurun_out/r2u/t/**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "gb_dict_clear" in n]
last = idx[-1]
prev_end = None
for r in rows[last - 3:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{r['Kernel_Name'][:50]:50s} dur {(e - s) / 1e3:9.1f} us   gap before {gap:8.1f} us")
    prev_end = e
PY
rm -rf $GRAFT_REPO_ROOT/$O/t
