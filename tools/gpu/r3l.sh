# round 3, call l: the fused multi-GPU join's receiver on six-byte tuples + 16-byte key loads: parity, the one-rank simulation, headline check
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_fused_join.py tests/test_gpu_multirank_one_gpu.py tests/test_gpu_rccl_multi.py tests/test_gpu_join_internals.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_join.py -m gpu -x -q -k "six_byte or speculative or prepared or xcd" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -4 >> $O/pytest.txt
cat $O/pytest.txt
python tools/sim_c4_fused.py 2>/dev/null | tail -6 > $O/sim_c4.txt
LIBGDF_AMD_LAB=1 GDF_JK_NO_P6=1 python tools/sim_c4_fused.py 2>/dev/null | tail -4 >> $O/sim_c4.txt
python tools/sim_c4_local.py 2>/dev/null | tail -4 >> $O/sim_c4.txt
cat $O/sim_c4.txt
for i in 1 2; do python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})" >> $O/bench.txt; done
python bench.py --force-distributed --strategy fused --steps 5 --warmup 2 --probe-rows 1000000000 --build-rows 125000000 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force-distributed fused', round(d['ms_per_step'],3), d['kernels_ms_per_step'])" >> $O/bench.txt
cat $O/bench.txt
