"""Where do jk_scatter1's two modes (3.25 / 3.65 ms per C3 join, stable inside a process, different between processes of one box) come from?
One process, several ROUNDS: in every round the probe / build columns are re-created (fresh torch allocations after empty_cache) and, in the
odd rounds, the RMM pool's scratch is shifted by a dummy block of a few MiB first.  Per round: kernel times of 5 joins."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd import api
from libgdf_amd.columns import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import make_probe_keys, make_build_keys, read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
dev = torch.device("cuda", 0)
npr, nb = 1_000_000_000, 100_000_000
for rnd in range(6):
    torch.cuda.empty_cache()
    pad = torch.empty((rnd * 37 + 1) << 20, dtype=torch.uint8, device=dev) if rnd >= 3 else None      # rounds 3..5: the columns land elsewhere
    build = make_build_keys(nb, 0x5EED0001, dev)
    probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
    dummy = C.c_void_p()
    if rnd & 1:
        gdf.librmm.rmmAlloc(C.byref(dummy), C.c_size_t((3 + rnd) * 1234567), None)
    for _ in range(2):
        li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    torch.cuda.synchronize()
    for _ in range(5):
        li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    torch.cuda.synchronize()
    lib.gdf_amd_profile_enable(0)
    prof = read_profile(gdf)
    print(json.dumps({"round": rnd, "probe_ptr": hex(probe.data_ptr()), "build_ptr": hex(build.data_ptr()), "scratch_shifted": bool(rnd & 1),
                      "ms": {k: round(v[0] / 5, 3) for k, v in prof.items() if v[0] / 5 > 0.1}}), flush=True)
    if dummy.value:
        gdf.librmm.rmmFree(dummy, None)
    del build, probe, pad
