"""Placed scratch blocks (librmm gdf_amd_rmm_place_*): one process, C3 joins one after another with the pool's decisions printed.
Per call: wall ms (includes the hipMalloc of a challenger while the pool is exploring); at the end the pool's trace and five profiled calls."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd import api
from libgdf_amd.columns import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib, _rmm_cdll as rmm
from bench import make_probe_keys, make_build_keys, read_profile
dev = torch.device("cuda", 0)
npr, nb = 1_000_000_000, 100_000_000
draws = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
rmm.gdf_amd_rmm_place_draws(C.c_int(draws))
build = make_build_keys(nb, 0x5EED0001, dev)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
walls = []
for i in range(draws + 4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    torch.cuda.synchronize(); walls.append(round((time.perf_counter() - t0) * 1e3, 2))
rmm.gdf_amd_rmm_place_trace.restype = C.c_size_t
n = rmm.gdf_amd_rmm_place_trace(None, C.c_size_t(0))
buf = C.create_string_buffer(n)
rmm.gdf_amd_rmm_place_trace(buf, C.c_size_t(n))
lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5 * 1e3
lib.gdf_amd_profile_enable(0)
prof = read_profile(gdf, split_sides=True)
print(json.dumps({"draws": draws, "first_calls_wall_ms": walls, "settled_ms_per_join": round(wall, 3),
                  "ms": {k: round(v[0] / 5, 3) for k, v in prof.items() if v[0] / 5 > 0.1}, "trace": buf.value.decode().splitlines()}), flush=True)
