import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import libgdf_amd as gdf
from libgdf_amd.columns import Column
g = torch.Generator(device="cuda"); g.manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
nb, npr, space, base = 1361026, 15198931, 200000000, -1000000
build = torch.randint(0, space, (nb,), generator=g, device="cuda")
probe = torch.randint(0, space, (npr,), generator=g, device="cuda")
probe[torch.randint(0, npr, (npr // 10,), generator=g, device="cuda")] = int(build[0])
bk, pk = (build + base).to(torch.int32), (probe + base).to(torch.int32)
mult = torch.bincount(build, minlength=space)
expected = int(mult[probe].sum())
li, ri = gdf.api.join([Column(pk)], [Column(bk)])
torch.cuda.synchronize()
print("ok", li.numel(), expected, flush=True)
