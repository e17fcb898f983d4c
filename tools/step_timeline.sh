#!/bin/bash
# Development aid (run through gpurun): kernel-by-kernel timeline (start offset, duration, gap to the previous kernel) of the LAST traced step
# of a bench workload.   usage: tools/step_timeline.sh "<bench.py arguments>"   -> gpurun_out/timeline.txt
ARGS=${@:-"--steps 3 --warmup 2"}
ARGS="$ARGS --no-secondary --cpu-pairs 0 --parity-pairs 0 --no-pmc --no-sequences"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tl && mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python bench.py $ARGS > /tmp/tl_bench.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last pyramid_fused_kernel pair backwards: find the start of the last keyframe pyramid (second to last pyramid launch)
pyr = [i for i, r in enumerate(rows) if "pyramid_fused_kernel" in r["Kernel_Name"]]
i0 = pyr[-2]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
out = []
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void vors::", "").replace("vors::", "")[:60]
    out.append(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {r.get('Grid_Size_X', '?'):>8}  {name}")
    prev_end = max(prev_end, e)
out.append(f"step total {(prev_end - t0) / 1e3:.1f} us, sum of gaps {sum(max(0, int(b['Start_Timestamp']) - int(a['End_Timestamp'])) for a, b in zip(rows[i0:], rows[i0 + 1:])) / 1e3:.1f} us")
open("gpurun_out/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[-70:]))
PY
