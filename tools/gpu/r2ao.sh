ulimit -c 0
mkdir -p gpurun_out/r2ao
timeout 200 python tools/stress_join.py --seed 21 --case 237 2>&1 | tail -1
timeout 200 python tools/stress_join.py --seed 22 --case 57 2>&1 | tail -1
timeout 200 python tools/stress_join.py --seed 23 --case 945 2>&1 | tail -1
for seed in 31 32 33 34; do
GDF_STRESS_VERBOSE=1 timeout 500 python tools/stress_join.py --seconds 240 --seed $seed > gpurun_out/r2ao/out_$seed.txt 2>&1
echo "seed $seed:"; grep "^case" gpurun_out/r2ao/out_$seed.txt | tail -1; tail -1 gpurun_out/r2ao/out_$seed.txt
done
GDF_STRESS_SECONDS=150 timeout 1200 python -m pytest tests/test_gpu_stress.py tests/test_gpu_join.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8
