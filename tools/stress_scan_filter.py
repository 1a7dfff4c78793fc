#!/usr/bin/env python
"""Randomised check of gdf_prefixsum_* and gpu_apply_stencil around the sizes where their kernels change (the element-wise kernels of
unaligned slices, the coalesced multi-pass kernels, the one-pass lockstep-rounds kernels from 2^22 rows on -- csrc/scan.hip, csrc/filter.hip)
against torch on the same device: cumsum in int64 cast back to the column's dtype (the reference's sums wrap in the input dtype,
src/scan.cu:11-76) and boolean indexing (a stable copy_if, streamcompactionops.cu:162-205).
Usage: python tools/stress_scan_filter.py [--seconds S] [--seed N]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-rows", type=int, default=60_000_000)
    a = ap.parse_args()
    import torch
    os.environ.setdefault("LIBGDF_AMD_TESTHOOK", "1")
    import libgdf_amd as gdf
    from libgdf_amd.columns import Column, mask_from_bools
    g = torch.Generator(device="cuda")
    r = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=g, device="cuda"))
    t0 = time.time()
    it = 0
    edges = [1 << 22, (1 << 22) + 4096, 4096 * 768, 4096 * 768 * 4, 4096 * 1024 * 3]
    while time.time() - t0 < a.seconds:
        g.manual_seed(a.seed * 1_000_003 + it)
        kind = r(0, 3)
        if kind == 0:
            n = r(1, 300_000)
        elif kind == 1:
            n = max(1, edges[r(0, len(edges))] + r(-5000, 5000))
        else:
            n = r(1 << 22, a.max_rows)
        off = [0, 0, 1, 3][r(0, 4)]                # a slice that starts inside an allocation is not 16-byte aligned
        dt = [torch.int8, torch.int32, torch.int64][r(0, 3)]
        tag = (it, n, off, str(dt).replace("torch.", ""))
        if os.environ.get("GDF_STRESS_VERBOSE"):
            print("case", tag, flush=True)
        lo, hi = (-100, 100) if dt == torch.int8 else (-(2**31), 2**31 - 1)
        whole = torch.randint(lo, hi, (n + off,), generator=g, device="cuda").to(dt)
        col = whole[off:off + n]
        inc = bool(r(0, 2))
        got = gdf.api.prefixsum(Column(col), inc)
        exp = torch.cumsum(col.long(), 0)
        if not inc:
            exp = exp - col.long()
        assert torch.equal(got, exp.to(dt)), (tag, "prefix sum", inc)
        # gpu_apply_stencil: the same column under a random stencil (and sometimes a stencil validity mask)
        keep_p = [0.0, 0.01, 0.1, 0.5, 0.97, 1.0][r(0, 6)]
        st_whole = (torch.rand(n + off, generator=g, device="cuda") < keep_p).to(torch.int8)
        st = st_whole[off:off + n]
        sv = None
        if r(0, 3) == 0:
            sv = torch.rand(n, generator=g, device="cuda") < 0.8
        stc = Column(st) if sv is None else Column(st, torch.from_numpy(mask_from_bools(sv.cpu().numpy())).cuda(), null_count=int(n - int(sv.sum())))
        out = gdf.api.apply_stencil(Column(col), stc)
        want = col[(st != 0) if sv is None else ((st != 0) & sv)]
        assert out.size == want.numel(), (tag, "apply_stencil size", keep_p, out.size, want.numel())
        assert torch.equal(out.data[:out.size], want), (tag, "apply_stencil rows", keep_p)
        it += 1
        del whole, col, got, exp, st_whole, st, out, want
    print(f"stress_scan_filter: {it} columns in {time.time() - t0:.0f} s, prefix sums and compactions equal torch's (seed {a.seed})")


if __name__ == "__main__":
    main()
