# round 3, third call: A/B of the claim reorder (prev = the commit before), parity of the touched kernels, materialisation
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
for i in 1 2 3; do
  LIBGDF_AMD_LAB=prev python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})" >> $O/ab.txt
  python bench.py --steps 10 --warmup 3 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})" >> $O/ab.txt
done
cat $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_join.py tests/test_gpu_join_internals.py tests/test_gpu_fused_join.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest_join.txt
cat $O/pytest_join.txt
python tools/bench_shapes.py --only c3_materialise_2_payload_cols > $O/bench_shapes.jsonl 2>$O/bench_shapes.err
cut -c1-900 $O/bench_shapes.jsonl; tail -3 $O/bench_shapes.err
