#!/usr/bin/env python
"""Registers / scratch / LDS of every kernel of one csrc/*.hip file, from the device assembly hipcc emits for gfx950 (no GPU
needed).  Usage: python tools/kernel_regs.py join.hip [substring]   -- run after touching a kernel that sits at a register limit
(jk_scatter1: 128 VGPRs at 1024 threads; a spill there is a vmcnt(0) wait per reload)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    csrc = os.path.join(ROOT, "libgdf_amd", "csrc")
    out = f"/tmp/{os.path.basename(src)}.s"
    flags = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"-I{ROOT}/include", f"-I{csrc}", "--offload-arch=gfx950",
             "-munsafe-fp-atomics", "--cuda-device-only", "-S"]
    if os.path.basename(src) == "join.hip":
        flags += ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"]
    flags += [a for a in sys.argv[3:]]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, os.path.join(csrc, src), "-o", out])
    text = open(out).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        # the metadata block lists fields alphabetically around .name; take the enclosing entry
        pass
    entries = re.split(r"\n  - \.agpr_count:", text)
    rows = []
    for e in entries[1:]:
        name = re.search(r"\.name:\s+(\S+)", e)
        if not name:
            continue
        get = lambda k: (re.search(r"\." + k + r":\s+(\d+)", e) or [None, "?"])[1]
        rows.append((name.group(1), get("vgpr_count"), get("vgpr_spill_count"), get("sgpr_count"), get("private_segment_fixed_size"),
                     get("group_segment_fixed_size")))
    demangle = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for (n, v, sp, sg, scratch, lds), d in zip(rows, demangle):
        short = re.sub(r"\(.*", "", d).replace("gdf_amd::", "").replace("void ", "")
        if want in short:
            print(f"{short:70s} vgpr {v:>4s} spill {sp:>3s} sgpr {sg:>4s} scratch {scratch:>5s} lds {lds}")


if __name__ == "__main__":
    main()
