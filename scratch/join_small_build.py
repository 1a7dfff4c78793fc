import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
npr = 500_000_000
for nb in (100, 3000, 100_000, 3_000_000, 30_000_000):
    b = torch.randperm(nb, device="cuda")
    p = torch.randint(0, nb, (npr,), device="cuda")
    fn = lambda: gdf.api.join([Column(p)], [Column(b)], copy=False)
    fn(); torch.cuda.synchronize()
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); w = (time.perf_counter() - t0) * 1e3
    lib.gdf_amd_profile_enable(0)
    top = sorted(read_profile(gdf).items(), key=lambda kv: -kv[1][0])[:4]
    print(f"5e8 x {nb:9d}: {w:8.2f} ms  pairs {r[0].numel()}  " + ", ".join(f"{k} {v[0]:.2f}" for k, v in top), flush=True)
    del r, b, p
