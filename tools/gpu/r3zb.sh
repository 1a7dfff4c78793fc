# round 3, call zb: hole filling as (tail, hole) segment copies; six-byte tuples' remainders mixed before the cuckoo slot hashes: parity, then shapes / fused simulation new vs old (HEAD)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3zb
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_join.py tests/test_gpu_fused_join.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest.txt
cat $O/pytest.txt
for v in new ab_old new ab_old; do
  if [ $v = new ]; then unset LIBGDF_AMD_LAB; else export LIBGDF_AMD_LAB=$v; fi
  python tools/bench_shapes.py --only c3_headline,c3_80pct_hit,c3_half_hit,c3_masked_99pct_valid 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v', d.get('shape'), round(d.get('ms',0),2), {k:round(v,2) for k,v in d.get('kernels_ms',{}).items() if v>0.05})" >> $O/shapes.txt
  echo "$v $(python tools/sim_c4_fused.py 2>/dev/null | tail -2 | tr '\n' ' ')" >> $O/sim.txt
done
unset LIBGDF_AMD_LAB
cat $O/shapes.txt $O/sim.txt
