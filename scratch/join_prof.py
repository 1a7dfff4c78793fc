"""Per-kernel profile of a few non-C3 join shapes."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
def prof(name, fn):
    fn(); torch.cuda.synchronize()
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); w = (time.perf_counter() - t0) * 1e3
    lib.gdf_amd_profile_enable(0)
    print(f"{name}: wall {w:.2f} ms", {k: (round(v[0], 2), v[1]) for k, v in read_profile(gdf).items() if v[0] > 0.03}, flush=True)
nb, npr = 20_000_000, 200_000_000
b = torch.randperm(nb, device="cuda"); p = torch.randint(0, nb, (npr,), device="cuda")
pn = p[:nb].clone()
prof("full 2e7 x 2e7", lambda: gdf.api.join([Column(pn)], [Column(b)], how="full", copy=False))
prof("left 2e7 x 2e7", lambda: gdf.api.join([Column(pn)], [Column(b)], how="left", copy=False))
pf, bf = p.double(), b.double()
prof("inner float64 2e8 x 2e7", lambda: gdf.api.join([Column(pf)], [Column(bf)], copy=False))
del pf, bf
p2, b2 = (p % 7).int(), (b % 7).int()
prof("inner (int64,int32) 2e8 x 2e7", lambda: gdf.api.join([Column(p), Column(p2)], [Column(b), Column(b2)], copy=False))
