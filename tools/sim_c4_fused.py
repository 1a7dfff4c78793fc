"""Local (non-network) cost of one rank's share of C4 at world = 8 through the FUSED path (libgdf_amd.multigpu.fused_inner_join,
gdf_amd_fj_*), simulated on one GPU: the sender's level-1 regroup of the build relation and of four probe slices with the
world-8 layout, and -- a receive buffer being `world` blocks in exactly the layout of a send buffer -- the sender's own
buffers fed back as the receive buffers, so that the receiver handles the volume eight senders would deliver (plus one
device copy per buffer, the stand-in for the receive, as in tools/sim_c4_local.py)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("LIBGDF_AMD_TESTHOOK", "1")      # path switches (--force ...) go through libgdf_testhook.so: loaded in front of libgdf.so
import libgdf_amd as gdf
from libgdf_amd import api
from libgdf_amd.columns import Column
from libgdf_amd._binding import rmmOptions_t, _gdf_cdll as lib
from bench import make_probe_keys, make_build_keys, read_profile
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
for sw in sys.argv[1:]:          # path switches, e.g. GDF_FJ_NO_POW2 (gdf_amd_debug_force)
    gdf.libgdf.gdf_amd_debug_force(sw.split("=")[0].encode(), (sw.split("=")[1] if "=" in sw else "1").encode())
dev = torch.device("cuda", 0)
W = 8
npr, nb = 1_000_000_000, 125_000_000
build = make_build_keys(nb, 0x5EED0001, dev)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
def sync(): torch.cuda.synchronize()
PHASES = True       # False: no synchronisation between the phases (the last iterations: what the C entry's own sequence looks like)
def timed(name, fn, acc):
    if not PHASES: return fn()
    sync(); t = time.perf_counter(); r = fn(); sync(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3; return r
chunks = 4; step = npr // chunks
lay_b = api.fj_plan(W, nb * W, nb)
lay_p = api.fj_plan(W, nb * W, step, npr / nb)
print("layout: fine bits", lay_b.fine_bits, "coarse bits", lay_b.coarse_bits, "bins", W << lay_b.coarse_bits, "cap build/probe", lay_b.cap, lay_p.cap,
      "slack probe %.3f" % (W * lay_p.block / step - 1.0))
for it in range(7):
    acc = {}
    PHASES = it < 4
    if it == 3:
        lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    sync(); t0 = time.perf_counter()
    lo, hi = (int(x) for x in torch.aminmax(build))
    bk, brows, bfill, over = timed("send (level 1)", lambda: api.fj_send(Column(build), lo, hi, lay_b, 0), acc)
    assert not over
    rbk = timed("recv stand-in", lambda: bk.clone(), acc)
    b = timed("build (level 2)", lambda: api.FjBuild(rbk, bfill, lo, lay_b, nb), acc)
    a = b.accumulate(npr)
    per_buf = W * lay_p.block
    for c in range(chunks):
        pk, prow, pfill, over = timed("send (level 1)", lambda: api.fj_send(Column(probe[c * step:(c + 1) * step]), lo, hi, lay_p, c * step), acc)
        assert not over
        rk = timed("recv stand-in", lambda: pk.clone(), acc)
        timed("probe (level 2 + LDS probe)", lambda: a.add_recv(rk, pfill, lay_p, c * per_buf), acc)
    li, ri = timed("probe (level 2 + LDS probe)", lambda: a.finish(copy=False), acc)
    total = li.numel()
    del li, ri
    b.close()
    sync(); wall = (time.perf_counter() - t0) * 1e3
    print("iter", it, "pairs", total, "wall ms %.1f" % wall, {k: round(v, 2) for k, v in acc.items()} if PHASES else "(phases not synchronised)", flush=True)
    if it == 3: lib.gdf_amd_profile_enable(0)
lib.gdf_amd_profile_enable(0)
print({k: (round(v[0], 3), v[1]) for k, v in read_profile(gdf).items()})
