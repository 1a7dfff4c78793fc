# round 3, call h: two-level gdf_hash_partition, strided range sample, fj bound checks: full suite + operator benches + C4 sims
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
python tools/bench_ops.py > $O/bench_ops.jsonl 2>$O/bench_ops.err
python - <<'PY'
import json
for l in open('gpurun_out/r3h/bench_ops.jsonl'):
    d=json.loads(l); print(d['op'], round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])
PY
python tools/bench_c5.py 2>/dev/null | tail -1 | cut -c1-600
python tools/sim_c4_fused.py 2>/dev/null | tail -8 > $O/sim_c4.txt
python tools/sim_c4_local.py 2>/dev/null | tail -8 >> $O/sim_c4.txt
cat $O/sim_c4.txt
