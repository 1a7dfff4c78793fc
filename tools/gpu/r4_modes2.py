"""jk_scatter1's two per-process modes, second experiment: does RE-ALLOCATING the library's scratch (rmmFinalize + rmmInitialize between rounds:
every cached block is hipFree'd and hipMalloc'ed again by the next join) move a process from one mode to the other?  The key columns stay."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd import api
from libgdf_amd.columns import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import make_probe_keys, make_build_keys, read_profile
dev = torch.device("cuda", 0)
npr, nb = 1_000_000_000, 100_000_000
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
build = make_build_keys(nb, 0x5EED0001, dev)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
hold = []
for rnd in range(8):
    if rnd:
        gdf.librmm.rmmFinalize()
        if rnd % 2 == 0:      # even rounds: some other allocation sits where the scratch was, the new scratch lands elsewhere
            hold.append(torch.empty((rnd * 613 + 211) << 20, dtype=torch.uint8, device=dev))
        gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
    for _ in range(2):
        li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    torch.cuda.synchronize()
    for _ in range(5):
        li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    torch.cuda.synchronize()
    lib.gdf_amd_profile_enable(0)
    prof = read_profile(gdf)
    print(json.dumps({"round": rnd, "held_MiB": sum(int(h.numel()) >> 20 for h in hold),
                      "ms": {k: round(v[0] / 5, 3) for k, v in prof.items() if v[0] / 5 > 0.1}}), flush=True)
