#!/usr/bin/env python3
"""Memory instructions, waits and branches of an instruction range of one kernel (indices as printed by tools/isa_loops.py):
    python tools/isa_memops.py file.s <kernel-name-substring> <first> <last>"""
import re
import sys

path, key, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.split(";")[0].strip().endswith(":"))
idx = 0
for l in lines[start + 1:]:
    t = l.split(";")[0].strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        if lo <= idx <= hi:
            print("      " + t)
        continue
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    if lo <= idx <= hi and (t.split()[0].startswith(("global_", "scratch_", "buffer_", "s_waitcnt", "s_cbranch", "s_branch", "s_barrier")) or "exec" in t):
        print(f"{idx:5d} {t[:90]}")
    idx += 1
    if idx > hi or t.startswith("s_endpgm"):
        break
