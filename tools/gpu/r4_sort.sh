#!/bin/bash
# hybrid sort: parity + timing
cd /root/repo
mkdir -p gpurun_out/r4s
timeout 900 python -m pytest tests/test_gpu_sort.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4s/pytest.txt
cat gpurun_out/r4s/pytest.txt
timeout 600 python tools/bench_ops.py --ops sort 2>&1 | tail -8 | tee gpurun_out/r4s/ops.txt
