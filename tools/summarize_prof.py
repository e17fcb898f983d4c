"""Condense rocprofv3 CSV output (kernel stats + PMC counter passes) into small summaries that get committed to profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
lines = []


def find(sub, pattern):
    f = glob.glob(os.path.join(out_dir, sub, "**", pattern), recursive=True)
    return f[0] if f else None


LM_STAGE = ("lm_track_kernel", "lm_split_eval_kernel", "lm_split_step_kernel")
ONCE_PER_STEP = ("dense_idepth_level1", "keyframe_sparse_kernel", "dso_rounds_kernel")  # kernels launched exactly once per bench step


def is_lm(name):
    return any(k in name for k in LM_STAGE)


stats = find("trace", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    once = [int(r["Calls"]) for r in rows if any(k in r["Name"] for k in ONCE_PER_STEP)]
    steps = max(once) if once else 0
    lm_total = sum(float(r["TotalDurationNs"]) for r in rows if is_lm(r["Name"]))
    if steps:
        lines.append(f"# LM stage ({' + '.join(k for k in LM_STAGE if any(k in r['Name'] for r in rows))}): "
                     f"{lm_total / steps / 1e6:.3f} ms of kernel time per step over {steps} steps\n")
if stats:
    lines.append(f"# rocprofv3 --kernel-trace --stats  ({tag})\n")
    lines.append("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(stats)):
        name = r["Name"].split("(")[0][:70]
        lines.append(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                     f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    lm = [(k, n, v) for k, (n, v) in agg.items() if is_lm(k)]
    if lm:
        once = [n for k, (n, v) in agg.items() if any(o in k for o in ONCE_PER_STEP)]
        steps = max(once) if once else 0
        lines.append(f"\n# LM stage {counter}: {sum(v for k, n, v in lm) / max(steps, 1):.1f} KiB (raw counter) per step over {steps} steps")
    lines.append(f"\n# rocprofv3 --pmc {counter} (separate pass; raw counter, unit KiB per rocprof; per-launch average)\n")
    lines.append("| kernel | launches | avg per launch | total |\n|---|---|---|---|")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {n} | {v/n:.1f} | {v:.1f} |")
for log in ("bench_trace.log",):
    p = os.path.join(out_dir, log)
    if os.path.exists(p):
        js = [l for l in open(p).read().splitlines() if l.startswith('{"metric')]
        last = js[-1] if js else ""
        if last.startswith("{"):
            lines.append("\n# bench.py JSON line of the traced run\n\n```json\n" + last + "\n```")
open(os.path.join(out_dir, f"summary_{tag}.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines)[:6000])
