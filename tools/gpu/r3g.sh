# round 3, call g: build payload in LDS (jk_probe_bp): parity + the materialisation shape; hole filling from a third
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_join.py tests/test_gpu_stress.py -m gpu -x -q -k "not full_size and not headline and not 2_to_the_29" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_join.txt
cat $O/pytest_join.txt
python tools/bench_shapes.py --only c3_headline,c3_half_hit,c3_materialise_2_payload_cols > $O/bench_shapes.jsonl 2>$O/bench_shapes.err
python - <<'PY'
import json
for l in open('gpurun_out/r3g/bench_shapes.jsonl'):
    d=json.loads(l); print(d['shape'], round(d['ms'],2), d['out_rows'], d['kernels_ms'])
PY
tail -3 $O/bench_shapes.err
