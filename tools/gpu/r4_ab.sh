# quick A/B evidence for one build: the GPU tests that touch the changed kernels, then C5 in three processes, the fused simulation and two headline processes
set -x
TAG=${1:-r4ab}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py tests/test_gpu_sort.py tests/test_gpu_hash_partition.py tests/test_gpu_fused_join.py tests/test_gpu_filter.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_subset.txt; cat $O/pytest_subset.txt
for i in 1 2 3; do python tools/bench_c5.py 2>/dev/null | tail -1 >> $O/bench_c5.jsonl; done
python tools/sim_c4_fused.py 2>/dev/null | tail -4 > $O/sim_c4_fused.txt
for i in 1 2; do python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 --extra 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': d['ms_per_step'], 'kernels_ms_per_step': d['kernels_ms_per_step']}))" >> $O/bench_spread.jsonl; done
python tools/bench_ops.py > $O/bench_ops.jsonl 2>/dev/null
cut -c1-420 $O/bench_c5.jsonl; cat $O/sim_c4_fused.txt $O/bench_spread.jsonl; cut -c1-200 $O/bench_ops.jsonl
