#!/bin/bash
# Per-kernel durations of the LM stage in the last bench step (run through gpurun). Development aid.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/d1 && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d1/trace -o bench -- python bench.py --pairs ${PAIRS:-4096} --steps 3 --warmup 1 --no-secondary --cpu-pairs 0 > gpurun_out/d1/bench.log 2>&1; tail -1 gpurun_out/d1/bench.log | cut -c1-160; python - <<PY
import csv,glob
f=glob.glob("gpurun_out/d1/trace/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
# last step: from the last mode-1 lm_track launch to the end
idx=[i for i,r in enumerate(rows) if "lm_track_kernel" in r["Kernel_Name"]]
start=idx[-3]
t0=int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    n=r["Kernel_Name"]; n=n[n.find("vors::")+6:][:34]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {n}  grid {r.get('Grid_Size','?')}")
PY
rm -rf gpurun_out/d1/trace
