"""Local cost of one rank's share of C4 with the broadcast strategy at world = 2, 4, 8: narrow both relations, join the
local 1e9-row probe shard against world x 1.25e8 gathered build keys (the gather itself is not simulated)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import multigpu
from libgdf_amd._binding import rmmOptions_t
from bench import make_probe_keys, make_build_keys
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
dev = torch.device("cuda", 0)
npr = 1_000_000_000
for W in (2, 4, 8):
    nb = 125_000_000 * W
    build = make_build_keys(nb, 0x5EED0001, dev)
    probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p32 = multigpu._device_narrow(probe, 0, nb - 1)
        b32 = multigpu._device_narrow(build[: nb // W], 0, nb - 1)       # a rank narrows its own shard only
        torch.cuda.synchronize(); t1 = time.perf_counter()
        b_all = multigpu._device_narrow(build, 0, nb - 1) if it == 0 else b_all
        torch.cuda.synchronize(); t2 = time.perf_counter()
        li, ri = multigpu._device_join_columns(p32, b_all)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        n = li.numel(); del li, ri
    print(f"world {W}: narrow {1e3 * (t1 - t0):.2f} ms, join 1e9 x {nb:.1e} {1e3 * (t3 - t2):.2f} ms, pairs {n}", flush=True)
    del build, probe, b_all, p32, b32
