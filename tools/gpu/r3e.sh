# round 3, call e: full GPU suite on the current build, the 8(d) micro-metrics at N = 1e9, and a knob sweep of the two regroup
# kernels with the LAB build (lib/lab: experiment knobs read from the environment)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
python tools/bench_ops.py --rows 1000000000 --ops partition,scan,filter --reps 3 > $O/bench_ops_1e9.jsonl 2>$O/bench_ops.err
python - <<'PY'
import json
for l in open('gpurun_out/r3e/bench_ops_1e9.jsonl'):
    d=json.loads(l); print(d['op'], round(d['ms'],2), round(d['frac_of_8TBps'],3), d['kernels_ms'])
PY
run() {  # label, env...
  label=$1; shift
  env LIBGDF_AMD_LAB=1 "$@" python bench.py --steps 8 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.1})" >> $O/sweep.txt
}
run base
run sc512 GDF_JK_SC_THREADS=512
run sc2_512 GDF_JK_SC2_THREADS=512
run chunk64k GDF_JK_CHUNK_ROWS=65536
run chunk256k GDF_JK_CHUNK_ROWS=262144
run chunk1m GDF_JK_CHUNK_ROWS=1048576
run b1_7 GDF_JK_B1=7
run base
cat $O/sweep.txt
