"""LAB: where jk_scatter1's time goes on C3.  Run as
    LIBGDF_AMD_LAB=1 GDF_JK_CLOCK=1 [GDF_JK_SDBG=1|4|8] python tools/scatter1_clock.py
The LAB library prints the cycles the first and the last wave of every workgroup spent in each phase of the level-1 scatter
(csrc/join.hip LAB_PHASE); with a store ablation (GDF_JK_SDBG) the join's RESULT is garbage and any error it ends in is ignored --
only the level-1 kernel's clock and duration are of interest."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from libgdf_amd.columns import Column
from bench import make_probe_keys, make_build_keys, read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
dev = torch.device("cuda", 0)
npr, nb = 1_000_000_000, 100_000_000
build = make_build_keys(nb, 0x5EED0001, dev)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
for it in range(2):
    if it == 1:
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    try:
        li, ri = gdf.api.join([Column(probe)], [Column(build)], how="inner")
        print("pairs", li.numel(), flush=True)
        del li, ri
    except Exception as e:       # (ablations)
        print("join ended with", type(e).__name__, e, flush=True)
    torch.cuda.synchronize()
lib.gdf_amd_profile_enable(0)
print({k: (round(v[0], 3), v[1]) for k, v in read_profile(gdf).items() if v[0] > 0.2})
