"""Round 6: the first calls of a process, one C3 join after another: wall ms per call and the pool's trace (slow hipMalloc / hipFree calls are
trace lines of their own).  argv[1]: calls (default 8); argv[2]: "wide" for 64-bit keys."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd import api
from libgdf_amd.columns import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib, _rmm_cdll as rmm
from bench import make_probe_keys, make_build_keys, wide_unique_keys
dev = torch.device("cuda", 0)
npr, nb = 1_000_000_000, 100_000_000
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
if len(sys.argv) > 2 and sys.argv[2] == "wide":
    build = wide_unique_keys(nb, 0x5EED0031, dev)
    probe = build[make_probe_keys(npr, nb, 0x5EED0032, dev)]
else:
    build = make_build_keys(nb, 0x5EED0001, dev)
    probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
walls = []
for i in range(calls):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    li, ri = api.join([Column(probe)], [Column(build)], how="inner", copy=False); del li, ri
    torch.cuda.synchronize(); walls.append(round((time.perf_counter() - t0) * 1e3, 2))
rmm.gdf_amd_rmm_place_trace.restype = C.c_size_t
n = rmm.gdf_amd_rmm_place_trace(None, C.c_size_t(0))
buf = C.create_string_buffer(n)
rmm.gdf_amd_rmm_place_trace(buf, C.c_size_t(n))
st = (C.c_ulonglong * 4)()
rmm.gdf_amd_rmm_place_stats(st)
print(json.dumps({"calls_wall_ms": walls, "stats": list(st), "trace": buf.value.decode().splitlines()}), flush=True)
