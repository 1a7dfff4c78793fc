ulimit -c 0
mkdir -p gpurun_out/r2as
for seed in 41 42; do
GDF_STRESS_VERBOSE=1 timeout 700 python tools/stress_join.py --seconds 300 --seed $seed --max-build 300000000 --max-probe 600000000 > gpurun_out/r2as/out_$seed.txt 2>&1
echo "seed $seed:"; grep -c "^case" gpurun_out/r2as/out_$seed.txt; grep "^case" gpurun_out/r2as/out_$seed.txt | tail -1; tail -1 gpurun_out/r2as/out_$seed.txt
done
