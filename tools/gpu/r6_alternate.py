"""Round 6: TWO shapes alternating in one process -- the narrow C3 join and the same join on 64-bit keys, call after call -- wall ms per call and the
pool's statistics: do the placed blocks of two shapes fit the pool's entries, or does every call of one shape evict the other's?"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd import api
from libgdf_amd.columns import Column
from libgdf_amd._binding import rmmOptions_t, _rmm_cdll as rmm
from bench import make_probe_keys, make_build_keys, wide_unique_keys
dev = torch.device("cuda", 0)
npr, nb = 1_000_000_000, 100_000_000
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
wb = wide_unique_keys(nb, 0x5EED0031, dev)
wp = wb[make_probe_keys(npr, nb, 0x5EED0032, dev)]
nbk = make_build_keys(nb, 0x5EED0001, dev)
npk = make_probe_keys(npr, nb, 0x5EED0002, dev)
walls = {"narrow": [], "wide": []}
for i in range(rounds):
    for name, p, b in (("narrow", npk, nbk), ("wide", wp, wb)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        li, ri = api.join([Column(p)], [Column(b)], how="inner", copy=False); del li, ri
        torch.cuda.synchronize(); walls[name].append(round((time.perf_counter() - t0) * 1e3, 2))
st = (C.c_ulonglong * 4)()
rmm.gdf_amd_rmm_place_stats(st)
print(json.dumps({"alternating_calls_wall_ms": walls, "place_stats": list(st)}), flush=True)
# ... and each shape on its own, in the same process, afterwards
for name, p, b in (("narrow", npk, nbk), ("wide", wp, wb)):
    w = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        li, ri = api.join([Column(p)], [Column(b)], how="inner", copy=False); del li, ri
        torch.cuda.synchronize(); w.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(json.dumps({"alone_afterwards": name, "calls_wall_ms": w}), flush=True)
