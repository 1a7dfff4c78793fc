O=gpurun_out/r2l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hash_partition.py -x -q -m gpu -k "shuffle" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_multirank_one_gpu.py tests/test_gpu_join.py -x -q -m gpu > $O/pytest2.txt 2>&1; tail -3 $O/pytest2.txt
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | cut -c1-100,600-1500
timeout 300 python bench.py --force-distributed --strategy shuffle --steps 3 --warmup 1 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r['kernels_ms_per_step'])"
bash tools/gpu/gaps.sh 2>&1 | grep "make_units\|total gap"
