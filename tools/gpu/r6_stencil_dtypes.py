"""Round 6: gpu_apply_stencil of 1e9 rows by element width -- one pass in lockstep rounds (the default from 2^22 rows) against the two passes
(GDF_FL_NO_ROUNDS), 10 % and 50 % kept."""
import os, sys, time, json
os.environ["LIBGDF_AMD_TESTHOOK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd.columns import Column
n = 1_000_000_000
for dt, w in ((torch.int8, 1), (torch.int16, 2), (torch.int32, 4), (torch.int64, 8)):
    a = torch.randint(-100, 100, (n,), device="cuda", dtype=dt)
    for keep in (0.1, 0.5):
        st = (torch.rand(n, device="cuda") < keep).to(torch.int8)
        for mode in (None, "1", None, "1"):
            gdf.libgdf.gdf_amd_debug_force(b"GDF_FL_NO_ROUNDS", mode.encode() if mode else None)
            r = gdf.api.apply_stencil(Column(a), Column(st)); kept = r.size; del r
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                r = gdf.api.apply_stencil(Column(a), Column(st)); del r
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
            print(json.dumps({"dtype": str(dt).replace("torch.", ""), "kept": keep, "kernel": "rounds" if mode is None else "two passes", "ms": round(ms, 3),
                              "frac_of_8TBps": round(((w + 1.0) * n + w * kept) / (ms * 1e-3) / 8e12, 3)}), flush=True)
        del st
    del a
