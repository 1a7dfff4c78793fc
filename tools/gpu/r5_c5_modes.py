"""Does C5's scatter kernel have placement modes like jk_scatter1?  One process, the pool re-allocated every round (rmmFinalize +
rmmInitialize: every cached block hipFree'd and hipMalloc'ed again), per round the kernel times of three gdf_group_by_avg calls."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch
import libgdf_amd as gdf
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from libgdf_amd.columns import Column, column_array, new_context
from bench import read_profile
from bench_c5 import make_c5
dev = torch.device("cuda", 0)
n = 1_000_000_000
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
k0, k1, v, ok, mask = make_c5(n, dev)
kc = [Column(k0), Column(k1)]
vc = Column(v, mask, null_count=int(n - ok.sum().item()))
cap = 20_000_000
def out_col(tdtype, gdtype):
    return Column(torch.empty(cap, dtype=tdtype, device=dev), torch.zeros((cap + 7) // 8 + 64, dtype=torch.uint8, device=dev), gdtype, size=cap)
ok0, ok1, oagg = out_col(torch.int64, 4), out_col(torch.int32, 3), out_col(torch.float64, 6)
ka, oa = column_array(kc), column_array([ok0, ok1])
ctx = new_context(method=1)
call = lambda: gdf.libgdf.gdf_group_by_avg(2, ka, vc.ptr, None, oa, oagg.ptr, C.byref(ctx))
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    if rnd:
        gdf.librmm.rmmFinalize()
        gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
    call()
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        call()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
    lib.gdf_amd_profile_enable(0)
    prof = read_profile(gdf)
    print(json.dumps({"round": rnd, "ms": round(wall, 3), "kernels_ms": {k: round(x[0] / 3, 3) for k, x in prof.items() if x[0] / 3 > 0.1}}), flush=True)
