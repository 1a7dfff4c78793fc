import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import libgdf_amd as gdf
from libgdf_amd import multigpu
from libgdf_amd.columns import Column
from bench import make_probe_keys, make_build_keys
dev = torch.device("cuda", 0)
for npr, nb in ((100_000_000, 12_500_000), (1_000_000_000, 125_000_000)):
    build = make_build_keys(nb, 0x5EED0001, dev)
    probe = make_probe_keys(npr, nb, 0x5EED0002, dev)
    for chunks in (1, 4):
        r = multigpu.distributed_inner_join(probe, build, chunks=chunks)
        print(npr, nb, "chunks", chunks, "numel", r.numel(), [int(p.numel()) for p in r.probe_pos], flush=True)
    step = npr // 4
    for c in range(4):
        li, ri = gdf.api.join([Column(probe[c * step:(c + 1) * step])], [Column(build)])
        print("  direct join chunk", c, li.numel(), flush=True)
    outs, offs = gdf.api.hash_partition([Column(probe[:step]), Column(torch.arange(step, dtype=torch.int32, device=dev))], [0], 1)
    print("  partition P=1 equal multiset:", bool((torch.sort(outs[0].data).values == torch.sort(probe[:step]).values).all()), offs)
    del build, probe
