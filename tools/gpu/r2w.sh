O=gpurun_out/r2w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank_one_gpu.py tests/test_gpu_fused_join.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python tools/sim_c4_fused.py 2>>$O/err.txt | tail -4
timeout 600 python bench.py --force-distributed --strategy fused --steps 3 --warmup 1 --probe-rows 1000000000 --build-rows 125000000 2>>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['strategy'], round(d['ms_per_step'],2), d['config']['preflight'], {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})"
tail -3 $O/err.txt
