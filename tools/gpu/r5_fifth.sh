set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_e
mkdir -p $O
cd $R
for i in 1 2 3; do
  python tools/gpu/r5_place.py 4 > $O/place_trace_$i.json 2>$O/place_trace_$i.err; python -c "
import json; d=json.load(open('$O/place_trace_$i.json')); print(d['first_calls_wall_ms'], d['settled_ms_per_join'], d['ms']); print('\n'.join(l for l in d['trace'] if 'ms -1.000' not in l))"
done
B="python bench.py --steps 10 --warmup 6 --cpu-sample 0 --pandas-sample 0 --extra 0"
for i in 1 2 3 4; do
  for d in 0 4; do
    $B --place-draws $d 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'draws': $d, 'ms_per_step': d['ms_per_step'], 'kernels_ms_per_step': d['kernels_ms_per_step'], 'placement': d.get('placement')}))" >> $O/place_ab.jsonl
  done
done
cut -c1-330 $O/place_ab.jsonl
timeout 1500 python -m pytest tests/test_gpu_join.py tests/test_gpu_c5.py tests/test_gpu_rmm.py tests/test_gpu_join_internals.py -m gpu -q -x --durations=8 > $O/pytest_subset.txt 2>&1; tail -16 $O/pytest_subset.txt
