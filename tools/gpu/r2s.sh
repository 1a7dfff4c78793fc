O=gpurun_out/r2s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank_one_gpu.py tests/test_gpu_fused_join.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for s in auto fused shuffle broadcast; do
  timeout 600 python bench.py --force-distributed --strategy $s --steps 3 --warmup 1 --probe-rows 1000000000 --build-rows 125000000 2>>$O/err.txt | tail -1 > $O/bench_dist_$s.json
  python - "$O/bench_dist_$s.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print(d["config"]["strategy"], d["config"].get("strategy_planned"), round(d["ms_per_step"],2), d["config"]["preflight"], {k:round(v,2) for k,v in d["kernels_ms_per_step"].items()})
except Exception as e: print("parse fail", e, open(sys.argv[1]).read()[:300])
PY
done
for t in 256 512 1024; do echo "SC2 threads $t"; GDF_JK_SC2_THREADS=$t timeout 600 python tools/sim_c4_fused.py 2>>$O/err.txt | tail -3; done
tail -5 $O/err.txt
