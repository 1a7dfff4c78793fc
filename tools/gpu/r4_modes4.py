"""jk_scatter1's modes, fourth experiment: the scratch re-allocated every round (r4_modes2.py), ALTERNATING between plain hipMalloc blocks and
physically contiguous ones (librmm's gdf_amd_rmm_contiguous hook).  Same process, same virtual layout apart from what the runtime decides."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd import api
from libgdf_amd.columns import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib, _rmm_cdll
from bench import make_probe_keys, make_build_keys, read_profile
dev = torch.device("cuda", 0)
npr, nb = 1_000_000_000, 100_000_000
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_rmm_cdll.gdf_amd_rmm_contiguous.argtypes = [C.c_int]
_rmm_cdll.gdf_amd_rmm_contiguous.restype = None
_rmm_cdll.gdf_amd_rmm_contiguous(first)
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
build = make_build_keys(nb, 0x5EED0001, dev)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
for rnd in range(8):
    contiguous = (rnd + first) & 1
    if rnd:
        gdf.librmm.rmmFinalize()
        _rmm_cdll.gdf_amd_rmm_contiguous(contiguous)
        gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    torch.cuda.synchronize()
    first_call_ms = (time.perf_counter() - t0) * 1e3
    li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    lib.gdf_amd_profile_enable(0)
    prof = read_profile(gdf)
    print(json.dumps({"round": rnd, "contiguous": contiguous, "ms_per_join": round(wall, 3), "first_call_ms": round(first_call_ms, 1),
                      "ms": {k: round(v[0] / 5, 3) for k, v in prof.items() if v[0] / 5 > 0.3}}), flush=True)
