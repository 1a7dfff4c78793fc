O=gpurun_out/r2f; mkdir -p $O
for d in 0 128 32 64; do echo wide dbg=$d; GDF_JK_DBG=$d timeout 600 python tools/bench_shapes.py --only c3_wide_keys --reps 2 2>>$O/err.txt | cut -c280-900; done
echo nofastwide; GDF_JK_NO_FAST_WIDE=1 timeout 600 python tools/bench_shapes.py --only c3_wide_keys --reps 2 2>>$O/err.txt | cut -c280-900
echo dist-shuffle; timeout 600 python bench.py --force-distributed --strategy shuffle --steps 3 --warmup 1 --cpu-sample 0 --pandas-sample 0 2>>$O/err.txt | cut -c1-1500
tail -5 $O/err.txt
