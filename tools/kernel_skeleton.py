#!/usr/bin/env python
"""Memory / LDS / wait / branch skeleton of one kernel from the device assembly tools/kernel_regs.py leaves in /tmp (run that first):
which loads are in flight across which waits, where a branch cut a straight-line phase.  Usage:
python tools/kernel_skeleton.py groupby.hip <mangled-name-substring> [first [last]]"""
import re
import sys


def main():
    text = open(f"/tmp/{sys.argv[1]}.s").read()
    labels = [m.group(1) for m in re.finditer(r"^(_Z\S+):", text, re.M) if sys.argv[2] in m.group(1)]
    if not labels:
        sys.exit("no kernel matches")
    name = labels[0]
    i = text.index(name + ":")
    body = text[i:text.index("s_endpgm", i)]
    pat = re.compile(r"(global_load\w+|global_store\w+|global_atomic\w+|ds_\w+|s_waitcnt[^\n;]*|s_cbranch\w+ \S+|s_branch \S+|\.LBB\S+:|s_barrier|scratch_\w+|v_mov_b32)")
    seq = [m.group(1).strip() for line in body.splitlines() for m in [pat.match(line.strip())] if m]
    out, prev, c = [], None, 0
    for x in seq + [None]:
        if x == prev:
            c += 1
        else:
            if prev:
                out.append(f"{prev} x{c}" if c > 1 else prev)
            prev, c = x, 1
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(out)
    print(name, len(out))
    print("\n".join(out[lo:hi]))


if __name__ == "__main__":
    main()
