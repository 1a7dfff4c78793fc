import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
n = 100_000_000
def t(name, fn):
    fn(); torch.cuda.synchronize()
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); w = (time.perf_counter() - t0) * 1e3
    lib.gdf_amd_profile_enable(0)
    top = sorted(read_profile(gdf).items(), key=lambda kv: -kv[1][0])[:3]
    print(f"{name:60s} {w:8.2f} ms  groups {r[1].numel():8d}  " + ", ".join(f"{k} {v[0]:.2f}" for k, v in top), flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(5)
k32 = torch.randint(0, 10000, (n,), device="cuda", dtype=torch.int32, generator=g)
k64 = k32.long()
for vdt in (torch.int8, torch.int16, torch.int32, torch.float32, torch.int64, torch.float64):
    v = (torch.rand(n, device="cuda", generator=g) * 100).to(vdt)
    for op in ("sum", "avg"):
        t(f"{op} int32 keys 1e4 groups, {str(vdt)[6:]} values", lambda: gdf.api.group_by(op, [Column(k32)], Column(v), capacity=1 << 20))
k6 = torch.randint(0, 1_000_000, (n,), device="cuda", dtype=torch.int32, generator=g)
v32 = (torch.rand(n, device="cuda", generator=g) * 100).to(torch.int32)
t("sum int32 keys 1e6 groups, int32 values", lambda: gdf.api.group_by("sum", [Column(k6)], Column(v32), capacity=1 << 21))
