import os, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
for dtype in (torch.int64, torch.int32):
    for n in (1 << 20, 1 << 26, 1 << 27, (1 << 28) - 5, 1 << 28, 250_000_000, 1 << 29, 1_000_000_000):
        src = torch.arange(n, dtype=dtype, device=dev)
        for asyn in (False, True):
            dst = torch.full_like(src, -1)
            w = dist.all_to_all_single(dst, src, [n], [n], async_op=asyn)
            if asyn:
                w.wait()
            torch.cuda.synchronize()
            neq = (dst != src)
            bad = int(neq.sum().item())
            first = int(torch.nonzero(neq)[0].item()) if bad else -1
            print(dtype, n, "bytes", n * src.element_size(), "async", asyn, "mismatches", bad, "first", first, flush=True)
        del src, dst
