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


LM_STAGE = ("lm_track_kernel", "lm_split_eval_kernel", "lm_split_step_kernel", "lm_ref_track_kernel", "lm_ref_track_coop_kernel")
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
# Round 4: LM-stage kernels overlap (the side lane runs lm_track_kernel on a second stream under the level-0 rounds), so the SUM of kernel
# durations exceeds the stage's wall time. From the per-dispatch trace: per step, the span from the first LM-stage kernel's start to the last
# one's end, and the length of the union of their intervals — the figures HIP events around the stage (bench.py roofline.kernel_ms_avg) see.
trace = find("trace", "*kernel_trace.csv")
if trace:
    disp = []
    for r in csv.DictReader(open(trace)):
        disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    disp.sort()
    marks = [s for s, e, n in disp if any(k in n for k in ONCE_PER_STEP)]
    spans, unions = [], []
    for i, m in enumerate(marks):
        hi = marks[i + 1] if i + 1 < len(marks) else 1 << 62
        iv = [(s, e) for s, e, n in disp if m <= s < hi and is_lm(n)]
        if not iv:
            continue
        spans.append(max(e for s, e in iv) - min(s for s, e in iv))
        u, cs, ce = 0, iv[0][0], iv[0][1]
        for s, e in iv[1:]:
            if s > ce:
                u += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        unions.append(u + ce - cs)
    if spans:
        spans.sort(); unions.sort()
        lines.append(f"# LM stage wall time from the per-dispatch trace, median over {len(spans)} steps: first start -> last end {spans[len(spans) // 2] / 1e6:.3f} ms, "
                     f"union of the kernels' intervals {unions[len(unions) // 2] / 1e6:.3f} ms (kernels on two streams overlap: the sum above counts that time twice)\n")
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
# ---- machine-readable entry for profiles/lm_counters.json (bench.py reads it: roofline.traffic, valu_issue_frac)
import json
entry = {}
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    tot, steps_seen = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        if is_lm(r["Kernel_Name"]):
            tot += float(r["Counter_Value"])
        if any(o in r["Kernel_Name"] for o in ONCE_PER_STEP):
            steps_seen += 1
    if steps_seen:
        entry[counter.lower() + "_kib_raw_per_step"] = tot / steps_seen
f = find("pmc_sq", "*counter_collection.csv")
if f:
    agg, steps_seen = defaultdict(float), 0
    seen_disp = set()
    for r in csv.DictReader(open(f)):
        if is_lm(r["Kernel_Name"]):
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
        if any(o in r["Kernel_Name"] for o in ONCE_PER_STEP) and r["Counter_Name"] == "SQ_INSTS_VALU":
            steps_seen += 1
    if steps_seen:
        for k, v in agg.items():
            entry[k.lower()] = v / steps_seen
        lines.append(f"\n# SQ counters of the LM stage per step over {steps_seen} steps (own rocprofv3 --pmc pass): " +
                     ", ".join(f"{k} {v / steps_seen:.4g}" for k, v in sorted(agg.items())))
if "fetch_size_kib_raw_per_step" in entry and "write_size_kib_raw_per_step" in entry:
    # MI355X_MICROARCH.md: FETCH_SIZE (KiB) x 2 on gfx950, WRITE_SIZE (KiB) as is
    entry["traffic_bytes"] = int(entry["fetch_size_kib_raw_per_step"] * 1024 * 2 + entry["write_size_kib_raw_per_step"] * 1024)
entry["profile"] = f"profiles/{tag}_summary.md"
# the kernel sources this profile was taken from (bench.py kernel_source_hash: a committed counter entry is only quoted for the same sources)
import hashlib
_h = hashlib.sha256()
_src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visual-odometry-rs_amd", "csrc")
for _name in sorted(os.listdir(_src)):
    if _name.endswith((".hip", ".h", ".cpp")) or _name == "Makefile":
        _h.update(_name.encode())
        _h.update(open(os.path.join(_src, _name), "rb").read())
entry["source_sha16"] = _h.hexdigest()[:16]
open(os.path.join(out_dir, f"lm_counters_{tag}.json"), "w").write(json.dumps(entry, indent=1))
for log in ("bench_trace.log",):
    p = os.path.join(out_dir, log)
    if os.path.exists(p):
        js = [l for l in open(p).read().splitlines() if l.startswith('{"metric')]
        last = js[-1] if js else ""
        if last.startswith("{"):
            lines.append("\n# bench.py JSON line of the traced run\n\n```json\n" + last + "\n```")
open(os.path.join(out_dir, f"summary_{tag}.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines)[:6000])
