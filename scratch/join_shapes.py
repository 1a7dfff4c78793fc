"""Join shapes beyond C3, ms per call (pool allocator on): looks for pathologies, not for records."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import Column
from libgdf_amd._binding import rmmOptions_t
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
dev = "cuda"
def t(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    print(f"{name:70s} {(time.perf_counter() - t0) / reps * 1e3:8.2f} ms   pairs {r[0].numel()}", flush=True)
npr, nb = 200_000_000, 20_000_000
b = torch.randperm(nb, device=dev)
p = torch.randint(0, nb, (npr,), device=dev)
t("inner 2e8 x 2e7 int64 unique build", lambda: gdf.api.join([Column(p)], [Column(b)], copy=False))
pm = torch.where(p % 5 == 0, p + nb, p)
t("left  2e8 x 2e7 int64, 20% misses", lambda: gdf.api.join([Column(pm)], [Column(b)], how="left", copy=False))
p32, b32 = p.int(), b.int()
t("inner 2e8 x 2e7 int32", lambda: gdf.api.join([Column(p32)], [Column(b32)], copy=False))
pf, bf = p.double(), b.double()
t("inner 2e8 x 2e7 float64 keys", lambda: gdf.api.join([Column(pf)], [Column(bf)], copy=False))
p2, b2 = (p % 7).int(), (b % 7).int()
t("inner 2e8 x 2e7 (int64, int32) two-column keys", lambda: gdf.api.join([Column(p), Column(p2)], [Column(b), Column(b2)], copy=False))
pw, bw = p * (1 << 36), b * (1 << 36)
t("inner 2e8 x 2e7 int64 keys spread over 2^60 (wide tuples)", lambda: gdf.api.join([Column(pw)], [Column(bw)], copy=False))
t("inner 2e8 x 2e7 (int64 spread over 2^60, int32): hashed + verified", lambda: gdf.api.join([Column(pw), Column(p2)], [Column(bw), Column(b2)], copy=False))
del pm, p32, b32, pf, bf, pw, bw
bd = torch.randint(0, nb // 4, (nb,), device=dev)
pd = p[:50_000_000] % (nb // 4)
t("inner 5e7 x 2e7 build keys repeated 4x (200M pairs)", lambda: gdf.api.join([Column(pd)], [Column(bd)], copy=False))
z = (torch.rand(npr, device=dev) ** 6 * nb).long()
t("inner 2e8 x 2e7 skewed probe keys (u^6)", lambda: gdf.api.join([Column(z)], [Column(b)], copy=False))
pn = p[:nb].clone()
t("full  2e7 x 2e7 int64", lambda: gdf.api.join([Column(pn)], [Column(b)], how="full", copy=False))
