# final-state artifacts of a round -> gpurun_out/<tag> (the small summaries are copied into profiles/ afterwards)
# usage: bash tools/gpu/collect_profiles.sh r2z
set -x
TAG=${1:-r2z}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/pytest_gpu.txt
python bench.py 2>$O/bench.err | grep '^{' | tail -1 > $O/bench.json
python tools/bench_ops.py > $O/bench_ops.jsonl 2>/dev/null
python tools/bench_c5.py 2>/dev/null | tail -1 > $O/bench_c5.json
python tools/bench_shapes.py > $O/bench_shapes.jsonl 2>/dev/null
for s in fused shuffle broadcast; do
  python bench.py --force-distributed --strategy $s --steps 5 --warmup 2 --probe-rows 1000000000 --build-rows 125000000 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 > $O/bench_c4_local_$s.json
done
python tools/sim_c4_fused.py 2>/dev/null | tail -8 > $O/sim_c4.txt
python tools/sim_c4_local.py 2>/dev/null | tail -8 >> $O/sim_c4.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o join -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --pandas-sample 0 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o join -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pandas-sample 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o join -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pandas-sample 0 > $O/pmc_write.log 2>&1
# HBM bytes of the round's new kernels: C5's fused partition pass, the fused multi-GPU join, the LDS dictionary
pmc2() {   # tag, command...
  tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${tag}_f -o x -- "$@" > $O/pmc_${tag}.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${tag}_w -o x -- "$@" >> $O/pmc_${tag}.log 2>&1
  (cd $R && python tools/pmc_hbm_json.py $(find $O/pmc_${tag}_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_${tag}_w -name "*counter_collection.csv" | head -1) $O/pmc_${tag}_hbm_bytes.json "$*  (totals over all launches of the command: divide by launches)") > $O/pmc_${tag}.txt
}
pmc2 c5 python $R/tools/bench_c5.py --reps 1
if [ -z "$GDF_COLLECT_SHORT" ]; then      # (kernels untouched since the last full collection: GDF_COLLECT_SHORT=1 skips them)
pmc2 fused python $R/tools/sim_c4_fused.py
pmc2 c2sparse python $R/tools/bench_shapes.py --only c2_sparse_keys --reps 1
fi
cd $R
python tools/rocprof_summary.py $O/trace $O/kernel_stats.md
python tools/pmc_hbm_json.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm.json
# keep only the small summaries (the raw traces stay on the box)
rm -rf $O/trace/*/*.db; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*.db" -delete
cat $O/pytest_gpu.txt; cat $O/bench.json | cut -c1-600; du -sh $O
