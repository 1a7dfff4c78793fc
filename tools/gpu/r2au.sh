ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_join.py -x -q -m gpu -p no:cacheprovider -k "repeated_millions or global_table or oversize or skew" 2>&1 | tail -5
echo "seed 41 case 174"; GDF_STRESS_VERBOSE=1 timeout 400 python tools/stress_join.py --seed 41 --case 174 --max-build 300000000 --max-probe 600000000 2>&1 | grep -v amdgpu | tail -3
echo "seed 42 case 54"; GDF_STRESS_VERBOSE=1 timeout 400 python tools/stress_join.py --seed 42 --case 54 --max-build 300000000 --max-probe 600000000 2>&1 | grep -v amdgpu | tail -3
