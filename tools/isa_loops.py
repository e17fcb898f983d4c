#!/usr/bin/env python3
"""List the loops of one kernel in an AMDGPU assembly listing (hipcc -S --cuda-device-only) with a mnemonic histogram each:
    python tools/isa_loops.py file.s <kernel-name-substring> [min_instructions]
A loop = a backward branch to a label; nested loops are reported separately (outer counts include inner)."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_ins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i] and ".end_amdhsa_kernel" in "".join(lines[i:i + 400]))
for i in range(start, len(lines)):
    if lines[i].strip().startswith(".section") or lines[i].strip().startswith(".Lfunc_end"):
        end = i
        break
labels = {}
body = []
for i in range(start, end):
    l = lines[i].split(";")[0].rstrip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = len(body)
        continue
    t = l.strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    body.append(t)
print(f"kernel at line {start + 1}, {len(body)} instructions")
SLOW = ("v_cvt", "v_floor", "v_fract", "v_cmp", "v_cndmask", "v_bfe", "v_mad_u32", "v_mad_u64", "v_mad_i32", "v_addc", "v_rcp", "v_perm", "v_alignbit")
for j, ins in enumerate(body):
    m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", ins)
    if not m:
        continue
    tgt = labels.get(m.group(1) or m.group(2))
    if tgt is None or tgt > j or j - tgt < min_ins:
        continue
    seg = body[tgt:j + 1]
    h = collections.Counter(s.split()[0] for s in seg)
    valu = sum(c for k, c in h.items() if k.startswith("v_"))
    slow = sum(c for k, c in h.items() if k.startswith(SLOW) or "sdwa" in k)
    vmem = sum(c for k, c in h.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
    print(f"\nloop [{tgt}, {j}] {len(seg)} instr: VALU {valu} (slow-class {slow}), SALU {sum(c for k, c in h.items() if k.startswith('s_'))}, "
          f"VMEM {vmem}, LDS {sum(c for k, c in h.items() if k.startswith('ds_'))}")
    print("  " + ", ".join(f"{k} {c}" for k, c in h.most_common(60)))
