# round 6: the evidence set for the CURRENT kernel build (as tools/gpu/r4_collect.sh) -- PMC HBM bytes of one C3 join and one C5 group-by
# (two separate --pmc passes each; the placement search is OFF in these four runs: a single call would otherwise carry the calibration
# launches of DESIGN 3.9, and the traffic of a join does not depend on where its buffers lie), rocprofv3 kernel stats of the same commands
# (search on; calibration launches reported on their own lines), the headline line (roofline.traffic, extra.c2 / c5 / ops, three CPU
# baselines), five more headline processes, C5 in five processes (+ null keys, + per-workgroup segments), shapes, operators at 1e8 / 1e9
# rows, the fused multi-GPU simulation, the forced-distributed line, and the stress tools.
# usage: bash tools/gpu/r5_collect.sh <tag>            (GDF_COLLECT_PYTEST=1 also runs the GPU suite first)
set -x
TAG=${1:-r6z}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ -n "$GDF_COLLECT_PYTEST" ]; then timeout 3000 python -m pytest tests -m gpu -q --durations=40 > $O/pytest_gpu.txt 2>&1; tail -45 $O/pytest_gpu.txt; fi
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-sample 0 --pandas-sample 0 --extra 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o join -- $B --steps 1 --warmup 0 --place-draws 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o join -- $B --steps 1 --warmup 0 --place-draws 0 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o join -- $B --steps 3 --warmup 1 > $O/trace.log 2>&1
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
for i in 1 2 3 4 5; do python tools/bench_c5.py 2>/dev/null | tail -1 >> $O/bench_c5_five_processes.jsonl; done
python tools/bench_c5.py --null-keys 0.01 2>/dev/null | tail -1 > $O/bench_c5_nullkeys.json
for i in 1 2; do python tools/bench_c5.py --force GDF_GBP_NO_XCD 2>/dev/null | tail -1 >> $O/bench_c5_per_workgroup_segments.jsonl; done
python tools/bench_shapes.py > $O/bench_shapes.jsonl 2>/dev/null
for i in 1 2 3 4; do python tools/bench_shapes.py --only c3_wide_keys --reps 3 2>/dev/null | tail -1 >> $O/bench_wide_keys_four_processes.jsonl; done
for i in 1 2 3; do python tools/gpu/r6_place.py 8 > $O/place_trace_$i.json 2>/dev/null; done
python tools/gpu/r6_place.py 8 wide > $O/place_trace_wide.json 2>/dev/null
python tools/bench_ops.py > $O/bench_ops.jsonl 2>/dev/null
python tools/bench_ops.py --rows 1000000000 --ops partition,scan,filter > $O/bench_ops_1e9.jsonl 2>/dev/null
python tools/sim_c4_fused.py 2>/dev/null | tail -7 > $O/sim_c4_fused.txt
python bench.py --force-distributed --strategy fused --steps 5 --warmup 3 --probe-rows 1000000000 --build-rows 125000000 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 > $O/bench_force_distributed_fused.json
python tools/stress_join.py --seconds 120 --seed 11 > $O/stress_join.txt 2>&1; tail -3 $O/stress_join.txt
python tools/stress_groupby.py --seconds 90 --seed 12 > $O/stress_groupby.txt 2>&1; tail -3 $O/stress_groupby.txt
rm -rf $O/trace/*/*.db $O/trace_c5/*/*.db; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*.db" -delete
cut -c1-1100 $O/bench.json; cat $O/bench_spread.jsonl | cut -c1-200; cut -c1-300 $O/bench_c5_five_processes.jsonl; cat $O/pmc_hbm.txt $O/pmc_hbm_c5.txt; head -30 $O/kernel_stats.md; cut -c1-260 $O/bench_shapes.jsonl; cut -c1-220 $O/bench_ops_1e9.jsonl; cat $O/sim_c4_fused.txt; du -sh $O
