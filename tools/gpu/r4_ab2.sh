# A/B evidence for a build that touched the join / fused-join kernels: their GPU tests, the fused simulation, the forced-distributed line, headline processes
set -x
TAG=${1:-r4ab2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_join.py tests/test_gpu_fused_join.py tests/test_gpu_multirank_one_gpu.py tests/test_gpu_join_internals.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_subset.txt; cat $O/pytest_subset.txt
python tools/sim_c4_fused.py 2>/dev/null | tail -4 > $O/sim_c4_fused.txt
python tools/sim_c4_fused.py GDF_FJ_NO_POW2 2>/dev/null | tail -4 > $O/sim_c4_fused_nopow2.txt
python bench.py --force-distributed --strategy fused --steps 5 --warmup 2 --probe-rows 1000000000 --build-rows 125000000 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 > $O/bench_force_distributed_fused.json
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 --extra 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': d['ms_per_step'], 'kernels_ms_per_step': d['kernels_ms_per_step']}))" >> $O/bench_spread.jsonl; done
cat $O/sim_c4_fused.txt $O/sim_c4_fused_nopow2.txt $O/bench_spread.jsonl; cut -c1-400 $O/bench_force_distributed_fused.json
