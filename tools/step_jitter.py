"""step_jitter.py [steps] -- wall time of EVERY headline join call of one process (the call returns synchronised: its result sizes are read
back), after the same warm-up as bench.py: is a process's average a flat line or a few slow calls?  Prints the sorted times and the pool's
placement trace lines that fall into the timed calls."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import libgdf_amd as gdf
from libgdf_amd import gdf_column, libgdf, new_context
from libgdf_amd.columns import Column, column_array
from libgdf_amd._binding import rmmOptions_t
from bench import make_probe_keys, make_build_keys

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gdf.librmm.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
dev = torch.device("cuda", 0)
npr, nb = 1_000_000_000, 100_000_000
build = make_build_keys(nb, 0x5EED0001, dev)
probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
pcol, bcol = Column(probe), Column(build)
ctx = new_context()
la, ra = column_array([pcol]), column_array([bcol])
on = (C.c_int * 1)(0)
def step():
    li, ri = gdf_column(), gdf_column()
    libgdf.gdf_inner_join(la, 1, on, ra, 1, on, 1, 0, None, C.byref(li), C.byref(ri), C.byref(ctx))
    n = int(li.size)
    libgdf.gdf_column_free(C.byref(li)); libgdf.gdf_column_free(C.byref(ri))
    return n
for _ in range(5): step()
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    t = time.perf_counter(); step(); ts.append((time.perf_counter() - t) * 1e3)
print("calls in order:", " ".join("%.2f" % x for x in ts))
s = sorted(ts)
print("min %.3f median %.3f mean %.3f max %.3f" % (s[0], s[len(s) // 2], sum(s) / len(s), s[-1]))
