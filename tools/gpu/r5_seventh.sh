set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_fused_join.py tests/test_gpu_multirank_one_gpu.py tests/test_gpu_rmm.py tests/test_gpu_rccl_multi.py -m gpu -q -x --durations=6 > $O/pytest_fused.txt 2>&1; tail -14 $O/pytest_fused.txt
python tools/sim_c4_fused.py 2>/dev/null | tail -4 > $O/sim_c4_fused.txt; cat $O/sim_c4_fused.txt
python tools/sim_c4_fused.py GDF_FJ_NO_PREHASH 2>/dev/null | tail -4 > $O/sim_c4_fused_no_prehash.txt; cat $O/sim_c4_fused_no_prehash.txt
python bench.py --force-distributed --strategy fused --steps 5 --warmup 3 --probe-rows 1000000000 --build-rows 125000000 --cpu-sample 0 --pandas-sample 0 2>$O/fd.err | grep '^{' | tail -1 > $O/bench_force_distributed_fused.json; python -c "
import json; d=json.load(open('$O/bench_force_distributed_fused.json')); print(d['ms_per_step'], d['config'].get('nranks'), d['kernels_ms_per_step'])"
for i in 1 2 3; do python tools/bench_c5.py 2>/dev/null | tail -1 | cut -c1-420; done | tee $O/bench_c5.jsonl
timeout 1500 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py -m gpu -q -x 2>&1 | tail -4
