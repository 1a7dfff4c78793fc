"""Round 6: gdf_prefixsum_i64 at 1e9 rows -- the default (reduce-then-scan, 24 B per row of traffic) against the single-pass kernels the file
keeps (GDF_SCAN_LOOKBACK=1: decoupled look-back; 2: a spine workgroup), which lost at 1e8 rows in round 2 and had never been timed at 1e9."""
import os, sys, time, json
os.environ["LIBGDF_AMD_TESTHOOK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd.columns import Column
from bench import make_probe_keys
dev = torch.device("cuda", 0)
for n in (100_000_000, 1_000_000_000):
    vals = make_probe_keys(n, 1000, 0x5EED0004, dev)
    vc = Column(vals)
    want_last = int(vals.sum().item())
    for mode in (None, "1", "2", "3", None, "3"):
        gdf.libgdf.gdf_amd_debug_force(b"GDF_SCAN_LOOKBACK", mode.encode() if mode else None)
        r = gdf.api.prefixsum(vc, True); ok = int(r[-1].item()) == want_last; del r
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r = gdf.api.prefixsum(vc, True); del r
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        print(json.dumps({"rows": n, "GDF_SCAN_LOOKBACK": mode, "ms": round(ms, 3), "frac_of_8TBps": round(16.0 * n / (ms * 1e-3) / 8e12, 3), "ok": ok}), flush=True)
    del vals, vc
