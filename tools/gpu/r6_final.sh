# round 6, the FINAL evidence set (after the last change to csrc/): the GPU suite, PMC HBM bytes of one C3 join and one C5 group-by (separate
# --pmc passes, placement search off), rocprofv3 kernel stats, the driver-style headline line with roofline.traffic, five more headline processes.
# + the shapes (tools/bench_shapes.py), the operators at 1e9 rows and C5 in three processes on the same build
# usage: bash tools/gpu/r6_final.sh <tag>
set -x
TAG=${1:-r6final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.txt 2>&1; tail -30 $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-sample 0 --pandas-sample 0 --extra 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o join -- $B --steps 1 --warmup 0 --place-draws 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o join -- $B --steps 1 --warmup 0 --place-draws 0 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o join -- $B --steps 3 --warmup 4 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c5_fetch -o c5 -- python $R/tools/bench_c5.py --reps 1 --no-checks --place-draws 0 > $O/pmc_c5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c5_write -o c5 -- python $R/tools/bench_c5.py --reps 1 --no-checks --place-draws 0 > $O/pmc_c5_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o c5 -- python $R/tools/bench_c5.py --reps 3 --no-checks > $O/trace_c5.log 2>&1
cd $R
python tools/rocprof_summary.py $O/trace $O/kernel_stats.md
python tools/rocprof_summary.py $O/trace_c5 $O/kernel_stats_c5.md
python tools/pmc_hbm_json.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm.json "python bench.py --steps 1 --warmup 0 --place-draws 0 --cpu-sample 0 (one C3 join, placement search off)" > $O/pmc_hbm.txt
python tools/pmc_hbm_json.py $(find $O/pmc_c5_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_c5_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm_c5.json "python tools/bench_c5.py --reps 1 --no-checks --place-draws 0 (a warm-up call + one timed call: divide by launches)" > $O/pmc_hbm_c5.txt
cp $O/pmc_hbm.json $R/profiles/zz_tmp_pmc_hbm.json          # so that THIS run's bench line already carries the traffic
python bench.py --steps 20 --warmup 5 2>$O/bench.err | grep '^{' | tail -1 > $O/bench.json
rm -f $R/profiles/zz_tmp_pmc_hbm.json
for i in 1 2 3 4 5; do python bench.py --steps 10 --warmup 5 --cpu-sample 0 --pandas-sample 0 --extra 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': d['ms_per_step'], 'first_call_ms': d.get('first_call_ms'), 'warmup_calls_ms': d.get('warmup_calls_ms'), 'kernels_ms_per_step': d['kernels_ms_per_step'], 'placement': d.get('placement')}))" >> $O/bench_spread.jsonl; done
python tools/bench_shapes.py > $O/bench_shapes.jsonl 2>/dev/null
python tools/bench_ops.py --rows 1000000000 --ops partition,scan,filter > $O/bench_ops_1e9.jsonl 2>/dev/null
for i in 1 2 3; do python tools/bench_c5.py 2>/dev/null | tail -1 >> $O/bench_c5_three_processes.jsonl; done
rm -rf $O/trace/*/*.db $O/trace_c5/*/*.db; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*.db" -delete
cut -c1-900 $O/bench.json; cat $O/bench_spread.jsonl | cut -c1-220; cat $O/pmc_hbm.txt $O/pmc_hbm_c5.txt; head -24 $O/kernel_stats.md; cut -c1-260 $O/bench_shapes.jsonl; cut -c1-240 $O/bench_ops_1e9.jsonl; cut -c1-200 $O/bench_c5_three_processes.jsonl; du -sh $O
