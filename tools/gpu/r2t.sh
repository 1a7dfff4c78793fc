O=gpurun_out/r2t; mkdir -p $O
env | grep -i -E "nccl|rccl" 
cat > /tmp/pg.py <<'PY'
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
t=torch.ones(4,device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
print("JSONLINE", flush=True)
dist.destroy_process_group()
PY
echo "--- default"; python /tmp/pg.py 2>/dev/null | cat
echo "--- NCCL_DEBUG=WARN"; NCCL_DEBUG=WARN python /tmp/pg.py 2>/dev/null | cat
echo "--- NCCL_DEBUG=NONE"; NCCL_DEBUG=NONE python /tmp/pg.py 2>/dev/null | cat
echo "--- RCCL_MSCCL_ENABLE=0"; RCCL_MSCCL_ENABLE=0 python /tmp/pg.py 2>/dev/null | cat
echo "--- tests"
timeout 1500 python -m pytest tests/test_gpu_groupby.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python tools/bench_shapes.py --only c2_dense_keys,c2_sparse_keys 2>>$O/err.txt | cut -c1-400
GDF_GB_NO_LDS_DICT=1 timeout 600 python tools/bench_shapes.py --only c2_sparse_keys 2>>$O/err.txt | cut -c1-400
for s in auto fused shuffle broadcast; do
  timeout 600 python bench.py --force-distributed --strategy $s --steps 3 --warmup 1 --probe-rows 1000000000 --build-rows 125000000 2>>$O/err.txt | grep '^{' > $O/bench_dist_$s.json
  python - "$O/bench_dist_$s.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print(d["config"]["strategy"], d["config"].get("strategy_planned"), round(d["ms_per_step"],2), d["config"]["preflight"], {k:round(v,2) for k,v in d["kernels_ms_per_step"].items()})
except Exception as e: print("parse fail", e, open(sys.argv[1]).read()[:300])
PY
done
tail -5 $O/err.txt
