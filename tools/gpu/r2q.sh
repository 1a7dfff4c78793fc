O=gpurun_out/r2q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_join.py -x -q -m gpu -k "wide_keys" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python bench.py --force-distributed --strategy shuffle --steps 5 --warmup 2 --cpu-sample 0 --pandas-sample 0 2>$O/err.txt > $O/dist.json; python - <<'PY'
import json
r = json.loads(open("gpurun_out/r2q/dist.json").read().strip().splitlines()[-1])
print(r["ms_per_step"], {k: round(v, 3) for k, v in r["kernels_ms_per_step"].items()})
print(r.get("exchange"))
PY
timeout 300 python tools/sim_c4_local.py 2>>$O/err.txt | tail -5
