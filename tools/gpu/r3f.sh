# round 3, call f: hole-filling single pass (50 % .. 100 % hits), rmm window, C5 record-layout experiment (LAB knob)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_join.py tests/test_gpu_rmm.py tests/test_gpu_stress.py -m gpu -x -q -k "not full_size and not headline and not 2_to_the_29" 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest_join.txt
cat $O/pytest_join.txt
python tools/bench_shapes.py --only c3_headline,c3_masked_99pct_valid,c3_half_hit,c3_80pct_hit,c3_tenth_hit,c3_materialise_2_payload_cols > $O/bench_shapes.jsonl 2>$O/bench_shapes.err
python - <<'PY'
import json
for l in open('gpurun_out/r3f/bench_shapes.jsonl'):
    d=json.loads(l); print(d['shape'], round(d['ms'],2), d['out_rows'], d['kernels_ms'])
PY
for i in 1 2; do
LIBGDF_AMD_LAB=1 python tools/bench_c5.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('partition-major', round(d['ms'],2), d['kernels_ms'])" >> $O/c5_layout.txt
LIBGDF_AMD_LAB=1 GDF_GBP_CHUNK_MAJOR=1 python tools/bench_c5.py --reps 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunk-major (scatter only; the rest is the fallback path)', round(d['ms'],2), {k:v for k,v in d['kernels_ms'].items() if k.startswith('gbp')})" >> $O/c5_layout.txt
done
cat $O/c5_layout.txt
