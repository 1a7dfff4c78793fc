# the sparse single pass up to 55 % hits: join parity + stress, then the refresh of the build-stamped summaries (tools/gpu/r2zzz.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bq; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_fused_join.py tests/test_gpu_multirank_one_gpu.py tests/test_gpu_c5.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -10 > $O/pytest_join.txt
timeout 300 python tools/stress_join.py --seconds 90 --seed 505 > $O/stress505.txt 2>&1
cat $O/pytest_join.txt; tail -n 1 $O/stress505.txt
bash tools/gpu/r2zzz.sh r2zzzzz
