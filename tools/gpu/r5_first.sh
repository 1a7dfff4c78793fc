# round 5, first GPU call: (1) the placed-block pool A/B (draws 0 vs 4, alternating processes), (2) the GPU suite with --durations
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_a
mkdir -p $O
cd $R
B="python bench.py --steps 10 --warmup 6 --cpu-sample 0 --pandas-sample 0 --extra 0"
for i in 1 2 3 4; do
  for d in 0 4; do
    $B --place-draws $d 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'draws': $d, 'ms_per_step': d['ms_per_step'], 'kernels_ms_per_step': d['kernels_ms_per_step'], 'placement': d.get('placement'), 'probe_phase': (d['roofline'] or {}).get('probe_phase')}))" >> $O/place_ab.jsonl
  done
done
cat $O/place_ab.jsonl | cut -c1-400
python tools/gpu/r5_place.py 4 > $O/place_trace_4.json 2>$O/place_trace_4.err; tail -c 3000 $O/place_trace_4.json
python tools/gpu/r5_place.py 4 > $O/place_trace_4b.json 2>/dev/null
python bench.py --steps 5 --warmup 6 > $O/bench_full.json 2>$O/bench_full.err; cut -c1-1500 $O/bench_full.json
timeout 2400 python -m pytest tests -m gpu -q -x --durations=120 > $O/pytest_gpu.txt 2>&1; tail -140 $O/pytest_gpu.txt
