"""Partition / scan / filter / sort shapes beyond tools/bench_ops.py, ms per call: looks for pathologies."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
n = 100_000_000
def t(name, fn):
    fn(); torch.cuda.synchronize()
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); w = (time.perf_counter() - t0) * 1e3
    lib.gdf_amd_profile_enable(0)
    top = sorted(read_profile(gdf).items(), key=lambda kv: -kv[1][0])[:3]
    print(f"{name:64s} {w:8.2f} ms  " + ", ".join(f"{k} {v[0]:.2f}" for k, v in top), flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(9)
k64 = torch.randint(0, 1 << 40, (n,), device="cuda", generator=g)
k32 = torch.randint(0, 1 << 30, (n,), device="cuda", dtype=torch.int32, generator=g)
v8 = torch.randint(-100, 100, (n,), device="cuda", dtype=torch.int8, generator=g)
vf = torch.rand(n, device="cuda", dtype=torch.float64, generator=g)
mask = torch.randint(0, 256, ((n + 7) // 8 + 64,), device="cuda", dtype=torch.uint8, generator=g)
for P in (2, 64, 1000, 12000):
    t(f"hash_partition (int64 key, float64, int8) P={P}", lambda: gdf.api.hash_partition([Column(k64), Column(vf), Column(v8)], [0], P))
t("hash_partition (int32 key, int64) on 2 hash columns P=64", lambda: gdf.api.hash_partition([Column(k32), Column(k64)], [0, 1], 64))
t("hash_partition int64 key + masked float64 P=16", lambda: gdf.api.hash_partition([Column(k64), Column(vf, mask)], [0], 16, with_masks=True))
t("hash_rows (int32, int64, float64)", lambda: gdf.api.hash_rows([Column(k32), Column(k64), Column(vf)]))
for dt, col in (("int8", v8), ("int32", k32), ("int64", k64)):
    t(f"prefixsum {dt}", lambda: gdf.api.prefixsum(Column(col), True))
t("order_by (int32, int8) two columns", lambda: gdf.api.order_by([Column(k32), Column(v8)]))
t("order_by float64", lambda: gdf.api.order_by([Column(vf)]))
t("order_by int8", lambda: gdf.api.order_by([Column(v8)]))
