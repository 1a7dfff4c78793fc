ulimit -c 0
ulimit -v 200000000      # 200 GB of address space per process: a runaway host allocation fails instead of taking the box down
for t in multi_column fused partition_scan; do
  echo "== $t"
  GDF_STRESS_SECONDS=60 timeout 900 python -m pytest tests/test_gpu_stress.py -q -m gpu -p no:cacheprovider -x -k $t 2>&1 | grep -E "^E|passed|failed|Error" | head -20
done
