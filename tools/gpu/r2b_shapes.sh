set -x
O=gpurun_out/r2b; mkdir -p $O
python tools/bench_shapes.py > $O/shapes.jsonl 2> $O/shapes.err
for d in 0 1024; do GDF_JK_DBG=$d python bench.py --steps 10 --warmup 2 --cpu-sample 0 --pandas-sample 0 > $O/bench_dbg$d.json 2>> $O/bench.err; done
tail -c 600 $O/shapes.err
python - <<'PY'
import json
for d in (0, 1024):
    r = json.loads(open(f"gpurun_out/r2b/bench_dbg{d}.json").read().strip().splitlines()[-1])
    print(d, r["ms_per_step"], r["kernels_ms_per_step"])
for l in open("gpurun_out/r2b/shapes.jsonl"):
    r = json.loads(l); print(r["shape"], round(r["ms"], 2), round(r["frac_of_8TBps"], 3), r["kernels_ms"])
PY
