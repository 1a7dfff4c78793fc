#!/usr/bin/env python
"""VALU / SALU / LDS / VMEM instruction counts per basic block of one kernel (device assembly left in /tmp by tools/kernel_regs.py), with the
quarter-rate 32-bit multiplies counted apart: a wave64 VALU instruction occupies its SIMD16 for 4 cycles, v_mul_lo/hi_u32 for 16.
Usage: python tools/kernel_blocks.py join.hip <mangled-name-substring>"""
import re
import sys

text = open(f"/tmp/{sys.argv[1]}.s").read()
name = [m.group(1) for m in re.finditer(r"^(_Z\S+):", text, re.M) if sys.argv[2] in m.group(1)][0]
i = text.index(name + ":")
body = text[i:text.index("s_endpgm", i)]
lab, cnt, order = "entry", {}, ["entry"]
for line in body.splitlines():
    s = line.strip()
    m = re.match(r"(\.LBB\d+_\d+):", s)
    if m:
        lab = m.group(1)
        order.append(lab)
        continue
    if not s or s[0] in ";.":
        continue
    op = s.split()[0]
    c = cnt.setdefault(lab, dict(valu=0, mul32=0, salu=0, lds=0, vmem=0))
    if re.match(r"v_mul_(lo|hi)_[ui]32|v_mad_[ui]64_[ui]32", op):
        c["mul32"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.split("_")[0] in ("global", "scratch", "buffer", "flat"):
        c["vmem"] += 1
print(name)
tot = dict(valu=0, mul32=0, salu=0, lds=0, vmem=0)
for l in order:
    if l in cnt and sum(cnt[l].values()) > 8:
        print(f"{l:14s}", " ".join(f"{k} {v:4d}" for k, v in cnt[l].items()), f" simd-cycles {4 * cnt[l]['valu'] + 16 * cnt[l]['mul32']}")
    if l in cnt:
        for k in tot:
            tot[k] += cnt[l][k]
print("total", tot)
