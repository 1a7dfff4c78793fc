ulimit -c 0
ulimit -v 200000000
for t in multi_column group_by fused partition_scan; do
  echo "== $t"
  GDF_STRESS_SECONDS=240 timeout 1500 python -m pytest tests/test_gpu_stress.py -q -m gpu -p no:cacheprovider -x -k $t 2>&1 | grep -E "^E|passed|failed|Error" | head -25
done
