"""Local (non-network) cost of one rank's share of C4 at world=8, simulated on one GPU: fused narrow + row numbers +
partition P=8 per probe slice, the partitioned slice stands in for the received one (plus one device copy of it, the
stand-in for the receive), probe of the prepared build relation."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import multigpu
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import make_probe_keys, make_build_keys, read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
dev = torch.device("cuda", 0)
W = 8
npr, nb = 1_000_000_000, 125_000_000
build = make_build_keys(nb, 0x5EED0001, dev)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
def sync(): torch.cuda.synchronize()
def timed(name, fn, acc):
    sync(); t = time.perf_counter(); r = fn(); sync(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3; return r
old = "--old" in sys.argv
for it in range(4):
    acc = {}
    if it == 3:
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    sync(); t0 = time.perf_counter()
    lo, hi = (int(x) for x in torch.aminmax(build))
    bk, bp, boff = timed("shuffle", lambda: multigpu._device_shuffle(build, 0, W, (lo, hi)), acc)
    prepared = timed("prepare", lambda: multigpu._device_prepare(bk), acc)
    chunks = 4; step = npr // chunks; total = 0
    acc_obj = prepared.accumulate(npr)
    for c in range(chunks):
        pk, pp, off = timed("shuffle", lambda: multigpu._device_shuffle(probe[c * step:(c + 1) * step], c * step, W, (lo, hi)), acc)
        rk = timed("recv stand-in", lambda: pk.clone(), acc); rp = timed("recv stand-in", lambda: pp[0].clone(), acc)
        timed("probe", lambda: acc_obj.add([multigpu._as_column(rk)]), acc)
    li, ri = timed("probe", lambda: acc_obj.finish(copy=False), acc)
    total += li.numel()
    del li, ri
    prepared.close()
    sync(); wall = (time.perf_counter() - t0) * 1e3
    print("iter", it, "pairs", total, "wall ms %.1f" % wall, {k: round(v, 2) for k, v in acc.items()}, flush=True)
lib.gdf_amd_profile_enable(0)
print({k: (round(v[0], 3), v[1]) for k, v in read_profile(gdf).items()})
