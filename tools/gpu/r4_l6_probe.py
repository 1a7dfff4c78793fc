"""Which level-1 path does a C3-shaped join of `rows` probe rows take, and what do its kernels cost?  (round 4: six-byte level-1 tuples)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
os.environ.setdefault("LIBGDF_AMD_TESTHOOK", "1")      # path switches (--force ...) go through libgdf_testhook.so: loaded in front of libgdf.so
import libgdf_amd as gdf
from bench import make_build_keys, make_probe_keys, read_profile
from libgdf_amd._binding import rmmOptions_t
from libgdf_amd.columns import Column
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
lib = gdf._binding._gdf_cdll
dev = torch.device("cuda", 0)
for rows in [int(x) for x in sys.argv[1:]]:
    nb = rows // 10
    build = make_build_keys(nb, 0x5EED0001, dev)
    probe = make_probe_keys(rows, nb, 0x5EED0002, dev)
    for force in ([], ["GDF_JK_NO_L6"]):
        for f in force:
            gdf.libgdf.gdf_amd_debug_force(f.encode(), b"1")
        gdf.libgdf.gdf_amd_debug_force(b"GDF_JK_FORCE_FB", b"15")
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
        li, ri = gdf.api.join([Column(probe)], [Column(build)], how="inner")
        torch.cuda.synchronize()
        lib.gdf_amd_profile_enable(0)
        prof = read_profile(gdf)
        ok = bool(torch.equal(probe[li.long()], build[ri.long()])) and li.numel() == rows and torch.unique(li).numel() == rows
        print(json.dumps({"rows": rows, "forced": force, "pairs": int(li.numel()), "ok": ok, "kernels_ms": {k: round(v[0], 3) for k, v in prof.items() if v[0] > 0.02}}), flush=True)
        for f in force:
            gdf.libgdf.gdf_amd_debug_force(f.encode(), None)
        del li, ri
