# round 3, second call: payload-carrying materialisation -- parity, then the C3 shape
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_join.py -m gpu -x -q -k "carried or materialisation or random_values or masked_single" 2>&1 | tail -15 > $O/pytest_carry.txt
cat $O/pytest_carry.txt
python tools/bench_shapes.py --only c3_headline,c3_materialise_2_payload_cols > $O/bench_shapes.jsonl 2>$O/bench_shapes.err
cut -c1-900 $O/bench_shapes.jsonl; tail -5 $O/bench_shapes.err
