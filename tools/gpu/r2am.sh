ulimit -c 0
mkdir -p gpurun_out/r2am
for seed in 21 22 23; do
GDF_STRESS_VERBOSE=1 timeout 400 python tools/stress_join.py --seconds 150 --seed $seed > gpurun_out/r2am/out_$seed.txt 2>&1
echo "seed $seed:"; grep "^case" gpurun_out/r2am/out_$seed.txt | tail -1; tail -1 gpurun_out/r2am/out_$seed.txt
done
