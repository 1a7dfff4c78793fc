import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
n = 1_000_000_000
g = torch.Generator(device="cuda"); g.manual_seed(5)
v = torch.randint(0, 1000, (n,), device="cuda", generator=g)
for groups in (10_000, 1_000_000, 300_000_000):
    k = torch.randint(0, groups, (n,), device="cuda", generator=g)
    fn = lambda: gdf.api.group_by("sum", [Column(k)], Column(v), capacity=min(n, groups + 16))
    fn(); torch.cuda.synchronize()
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); w = (time.perf_counter() - t0) * 1e3
    lib.gdf_amd_profile_enable(0)
    top = sorted(read_profile(gdf).items(), key=lambda kv: -kv[1][0])[:4]
    print(f"1e9 rows, {groups:10d} key values: {w:8.2f} ms  groups {r[1].numel()}  " + ", ".join(f"{kk} {vv[0]:.2f}" for kk, vv in top), flush=True)
    del k, r
