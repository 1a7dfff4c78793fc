"""Group-by shapes beyond C2 / C5, ms per call: looks for pathologies."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
n = 100_000_000
def t(name, fn):
    fn(); torch.cuda.synchronize()
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); w = (time.perf_counter() - t0) * 1e3
    lib.gdf_amd_profile_enable(0)
    top = sorted(read_profile(gdf).items(), key=lambda kv: -kv[1][0])[:3]
    print(f"{name:66s} {w:8.2f} ms  groups {r[1].numel():9d}  " + ", ".join(f"{k} {v[0]:.2f}" for k, v in top), flush=True)
g = torch.Generator(device="cuda"); g.manual_seed(5)
def ri(hi, dtype=torch.int64): return torch.randint(0, hi, (n,), device="cuda", dtype=dtype, generator=g)
v64, vf = ri(1000), torch.rand(n, device="cuda", dtype=torch.float64, generator=g)
cap = dict(capacity=1 << 25)
k4 = ri(10_000)
t("sum   int64 keys 1e4 groups, int64 values", lambda: gdf.api.group_by("sum", [Column(k4)], Column(v64), **cap))
t("min   int64 keys 1e4 groups, float64 values", lambda: gdf.api.group_by("min", [Column(k4)], Column(vf), **cap))
t("count int32 keys 1e4 groups", lambda: gdf.api.group_by("count", [Column(k4.int())], Column(v64), **cap))
k32 = k4.int()
t("sum   int32 keys 1e4 groups, float64 values", lambda: gdf.api.group_by("sum", [Column(k32)], Column(vf), **cap))
ka, kb = ri(300), ri(300, torch.int32)
t("sum   (int64, int32) keys 9e4 groups", lambda: gdf.api.group_by("sum", [Column(ka), Column(kb)], Column(v64), **cap))
kf = (k4.double() * 0.5)
t("sum   float64 keys 1e4 groups", lambda: gdf.api.group_by("sum", [Column(kf)], Column(v64), **cap))
k6 = ri(1_000_000)
t("avg   int64 keys 1e6 groups, float64 values", lambda: gdf.api.group_by("avg", [Column(k6)], Column(vf), **cap))
kw = k6 * (1 << 38)
t("sum   int64 keys 1e6 groups spread over 2^58", lambda: gdf.api.group_by("sum", [Column(kw)], Column(v64), **cap))
kz = (torch.rand(n, device="cuda", generator=g) ** 8 * 5_000_000).long()
t("sum   int64 keys skewed (u^8 over 5e6)", lambda: gdf.api.group_by("sum", [Column(kz)], Column(v64), **cap))
k8 = ri(30_000_000)
t("sum   int64 keys 3e7 groups", lambda: gdf.api.group_by("sum", [Column(k8)], Column(v64), **cap))
kf6 = k6.double() * 0.25
t("sum   float64 keys 1e6 groups", lambda: gdf.api.group_by("sum", [Column(kf6)], Column(v64), **cap))
