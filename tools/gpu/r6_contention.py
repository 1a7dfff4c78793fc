"""Round 6: the lockstep-rounds kernels (gdf_prefixsum, gpu_apply_stencil) under CONTENTION -- run this script in two processes at once on one
GPU.  Their workgroups wait for one another and must all be resident; two such kernels from two processes can each hold part of the CUs.  Then a
poll lasts longer than a quarter of a second, the kernel bails out and the call starts over with the multi-pass kernels.  Every result is checked
against torch; the profile says how many calls took the fallback (scan_apply / compact_write launches next to scan_rounds / compact_rounds)."""
import json, os, sys, time
os.environ.setdefault("LIBGDF_AMD_TESTHOOK", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import libgdf_amd as gdf
from libgdf_amd.columns import Column
from bench import read_profile
lib = gdf._binding._gdf_cdll
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
g = torch.Generator(device="cuda"); g.manual_seed(os.getpid())
a = torch.randint(-1000, 1000, (n,), generator=g, device="cuda", dtype=torch.int64)
st = (torch.rand(n, generator=g, device="cuda") < 0.3).to(torch.int8)
exp_scan = torch.cumsum(a, 0)
exp_keep = a[st != 0]
lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
ms, bad = [], 0
for i in range(calls):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = gdf.api.prefixsum(Column(a), True)
    o = gdf.api.apply_stencil(Column(a), Column(st))
    torch.cuda.synchronize(); ms.append(round((time.perf_counter() - t0) * 1e3, 2))
    if not torch.equal(s, exp_scan) or o.size != exp_keep.numel() or not torch.equal(o.data[:o.size], exp_keep):
        bad += 1
    del s, o
lib.gdf_amd_profile_enable(0)
prof = read_profile(gdf)
launches = {k.split("@")[0]: int(v[1]) if len(v) > 1 else None for k, v in prof.items() if k.split("@")[0] in ("scan_rounds", "scan_apply", "compact_rounds", "compact_write")}
print(json.dumps({"pid": os.getpid(), "rows": n, "calls": calls, "wrong_results": bad, "calls_ms": ms, "launches": launches}), flush=True)
