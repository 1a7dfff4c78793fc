"""hs_local ablation: time gdf_order_by on 1e8 62-bit keys with parts of the bucket sort switched off (results are wrong then)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
os.environ.setdefault("LIBGDF_AMD_TESTHOOK", "1")      # path switches (--force ...) go through libgdf_testhook.so: loaded in front of libgdf.so
import libgdf_amd as gdf
from libgdf_amd.columns import Column
from bench import read_profile
n = 100_000_000
g = torch.Generator(device="cuda"); g.manual_seed(5)
k = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
col = Column(k)
lib = gdf._binding._gdf_cdll
for dbg in sys.argv[1:] or ["0", "1", "2", "4"]:
    gdf.libgdf.gdf_amd_debug_force(b"GDF_HS_DBG", dbg.encode())
    for _ in range(2): gdf.api.order_by([col])
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    for _ in range(3): gdf.api.order_by([col])
    lib.gdf_amd_profile_enable(0)
    p = read_profile(gdf)
    print(dbg, {k_: round(v[0] / v[1], 3) for k_, v in p.items() if k_.startswith("hs_")}, flush=True)
