O=gpurun_out/r2ai; mkdir -p $O
echo "--- shapes headline, stage clock"; GDF_JK_DBG=512 timeout 600 python tools/bench_shapes.py --only c3_headline --reps 2 2>&1 | grep -E "join\]|c3_headline" | tail -40 | cut -c1-200
echo "--- bench.py, stage clock"; GDF_JK_DBG=512 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --pandas-sample 0 2>&1 | grep -E "join\]" | tail -16
