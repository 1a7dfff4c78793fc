"""Sender-side shuffle kernels and gdf_hash_partition at small fan-outs, ms per kernel (library HIP-event profile)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import make_probe_keys, read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
keys = make_probe_keys(n, 1_000_000_000, 7, "cuda")
rows = torch.arange(n, dtype=torch.int32, device="cuda")
k32 = keys.to(torch.int32)
def prof(fn):
    fn(); torch.cuda.synchronize()
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    for _ in range(3): fn()
    torch.cuda.synchronize(); lib.gdf_amd_profile_enable(0)
    return {k: round(v[0] / 3, 3) for k, v in read_profile(gdf).items() if not k.startswith("scan")}
for P in (1, 2, 4, 8, 16):
    print("P", P, "shuffle narrow", prof(lambda: gdf.api.shuffle_partition(Column(keys), P, narrow=(0, 1_000_000_000))),
          "| hash_partition int32+int32", prof(lambda: gdf.api.hash_partition([Column(k32), Column(rows)], [0], P)), flush=True)
