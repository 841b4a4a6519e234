"""Instruction-class counts of one kernel of the emitted assembly, split at s_barrier (rough phase profile: how many VALU /
LDS / VMEM / MFMA instructions a wave issues between barriers).  usage: asm_segments.py <file.s> <mangled-name substring>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(sys.argv[2]), l))
body = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    body.append(t)


def cls(t):
    op = t.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    return "salu" if op.startswith("s_") else "other"


seg, cur = [], []
for t in body:
    cur.append(t)
    if t.startswith("s_barrier"):
        seg.append(cur)
        cur = []
seg.append(cur)
print(f"{len(body)} instructions, {len(seg)} segments")
for i, sg in enumerate(seg):
    c = Counter(cls(t) for t in sg)
    print(f"  seg {i:2d}: {len(sg):5d}  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))
