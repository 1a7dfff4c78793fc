O=gpurun_out/r2r; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_multirank_one_gpu.py tests/test_gpu_fused_join.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 600 python tools/sim_c4_fused.py 2>$O/err.txt | tail -7
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | cut -c1-100,600-1500
