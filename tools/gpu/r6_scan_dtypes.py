import os, sys, time, json
os.environ["LIBGDF_AMD_TESTHOOK"]="1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd.columns import Column
for dt, w in ((torch.int8,1),(torch.int32,4),(torch.int64,8)):
    n = 1_000_000_000
    a = torch.randint(-100, 100, (n,), device="cuda", dtype=dt)
    for mode in (None, "0"):
        gdf.libgdf.gdf_amd_debug_force(b"GDF_SCAN_LOOKBACK", mode.encode() if mode else None)
        r = gdf.api.prefixsum(Column(a), True); del r
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5):
            r = gdf.api.prefixsum(Column(a), True); del r
        torch.cuda.synchronize(); ms=(time.perf_counter()-t0)/5*1e3
        print(json.dumps({"dtype": str(dt).replace("torch.",""), "rows": n, "kernel": "rounds (default)" if mode is None else "three launches (GDF_SCAN_LOOKBACK=0)", "ms": round(ms,3), "frac_of_8TBps": round(2.0*w*n/(ms*1e-3)/8e12,3)}), flush=True)
    del a
