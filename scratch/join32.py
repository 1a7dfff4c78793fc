import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd._binding import rmmOptions_t
from libgdf_amd.columns import Column
from bench import make_probe_keys, make_build_keys, read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
lib = gdf._binding._gdf_cdll
dev = torch.device("cuda", 0)
nb, npr = 100_000_000, 1_000_000_000
build = make_build_keys(nb, 0x5EED0001, dev).to(torch.int32)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev).to(torch.int32)
pc, bc = Column(probe), Column(build)
for it in range(3):
    a, b = gdf.api.join([pc], [bc], copy=False)
lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(5):
    a, b = gdf.api.join([pc], [bc], copy=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
lib.gdf_amd_profile_enable(0)
print("int32 keys 1e9 x 1e8:", dt * 1e3, "ms", a.numel(), {k: round(v[0] / 5, 3) for k, v in read_profile(gdf).items()})
