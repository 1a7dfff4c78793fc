"""Round 6: gdf_hash_partition of 1e9 rows x 2 int64 columns, the pair / single-stage kernels (part_scatter_pairs_kernel, part_scatter_cols8_kernel) against the generic tile kernel
(GDF_HP_NO_PAIRS + GDF_HP_NO_COLS8), alternating in one process so that both see the same output columns."""
import os, sys, time, json
os.environ["LIBGDF_AMD_TESTHOOK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import numpy as np
import libgdf_amd as gdf
from libgdf_amd.columns import Column
from bench import make_probe_keys, read_profile
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
keys = make_probe_keys(n, 10000, 0x5EED0003, dev)
vals = make_probe_keys(n, 1000, 0x5EED0004, dev)
kc, vc = Column(keys), Column(vals)
lib = gdf._binding._gdf_cdll
for P in (256, 128, 64, 32):
    for mode in ("new", "generic", "new", "generic"):
        gdf.libgdf.gdf_amd_debug_force(b"GDF_HP_NO_PAIRS", None if mode == "new" else b"1")
        gdf.libgdf.gdf_amd_debug_force(b"GDF_HP_NO_COLS8", None if mode == "new" else b"1")
        cols, offs = gdf.api.hash_partition([kc, vc], [0], P)
        ok = int(cols[0].data.sum().item()) == int(keys.sum().item()) and int(cols[1].data.sum().item()) == int(vals.sum().item())
        del cols
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            r = gdf.api.hash_partition([kc, vc], [0], P); del r
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        lib.gdf_amd_profile_enable(0)
        prof = {k: round(v[0] / 3, 3) for k, v in read_profile(gdf).items() if v[0] / 3 > 0.05}
        print(json.dumps({"P": P, "kernel": mode, "ms": round(ms, 3), "frac_of_8TBps": round(40.0 * n / (ms * 1e-3) / 8e12, 3), "kernels_ms": prof, "ok": ok}), flush=True)
