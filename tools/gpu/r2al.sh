O=gpurun_out/r2al; mkdir -p $O
timeout 900 python tools/stress_join.py --seconds 200 --seed 11 2>&1 | tail -5
timeout 900 python tools/stress_join.py --seconds 200 --seed 12 2>&1 | tail -5
GDF_STRESS_SECONDS=120 timeout 1200 python -m pytest tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -15
