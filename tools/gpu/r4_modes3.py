"""jk_scatter1's modes, third experiment: the scratch is re-allocated every round (as in r4_modes2.py) with the RMM event log on -- which ADDRESSES did
the round's large blocks get?  Prints, per round, kernel times and (address, MiB) of the blocks above 1 GiB in allocation order."""
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
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, True)))
build = make_build_keys(nb, 0x5EED0001, dev)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
for rnd in range(8):
    if rnd:
        gdf.librmm.rmmFinalize()
        gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, True)))
    li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    n = gdf.librmm.rmmLogSize()
    buf = C.create_string_buffer(n + 1)
    gdf.librmm.rmmGetLog(buf, n + 1)
    big, seen = [], set()
    for line in buf.value.decode().splitlines()[1:]:
        f = line.split(",")
        if f[0] == "Alloc" and int(f[4]) >= (1 << 30) and f[2] not in seen:
            seen.add(f[2]); big.append((f[2], int(f[4]) >> 20))
    li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    torch.cuda.synchronize()
    for _ in range(4):
        li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    torch.cuda.synchronize()
    lib.gdf_amd_profile_enable(0)
    prof = read_profile(gdf)
    print(json.dumps({"round": rnd, "ms": {k: round(v[0] / 4, 3) for k, v in prof.items() if v[0] / 4 > 0.3}, "blocks": big,
                      "probe_ptr": hex(probe.data_ptr())}), flush=True)
