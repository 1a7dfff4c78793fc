# after the last kernel-source change of the round: refresh what bench.py's roofline.traffic reads (the PMC summary is stamped with the
# kernel build id), the kernel stats of the same command, the headline line and the small benches.  (The GPU suite ran on this build in
# tools/gpu/r2bj.sh: 1309 passed.)
TAG=${1:-r2zzz}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o join -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pandas-sample 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o join -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pandas-sample 0 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o join -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --pandas-sample 0 > $O/trace.log 2>&1
cd $R
python tools/rocprof_summary.py $O/trace $O/kernel_stats.md
python tools/pmc_hbm_json.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm.json
cp $O/pmc_hbm.json $R/profiles/zz_tmp_pmc_hbm.json          # so that THIS run's bench line already carries the traffic
python bench.py 2>$O/bench.err | grep '^{' | tail -1 > $O/bench.json
rm -f $R/profiles/zz_tmp_pmc_hbm.json
python tools/bench_c5.py 2>/dev/null | tail -1 > $O/bench_c5.json
python tools/bench_ops.py > $O/bench_ops.jsonl 2>/dev/null
python tools/bench_shapes.py > $O/bench_shapes.jsonl 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/readback_lab.hip -o /tmp/readback_lab 2>/dev/null && /tmp/readback_lab > $O/readback_lab.txt 2>&1
if [ -n "$GDF_REFRESH_PYTEST" ]; then timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | head -5 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt; fi
rm -rf $O/trace/*/*.db; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*.db" -delete
cut -c1-700 $O/bench.json; cut -c1-300 $O/bench_c5.json; cat $O/readback_lab.txt
