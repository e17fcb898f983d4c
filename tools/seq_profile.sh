#!/bin/bash
# rocprofv3 --kernel-trace --stats of the lock-step sequence workload alone (64 sequences x 40 frames, one candidates mode; run through gpurun).
# usage: tools/seq_profile.sh [c2f|dense|dso]      -> gpurun_out/prof_seq_<mode>/kernel_stats.md
MODE=${1:-c2f}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_seq_$MODE
mkdir -p $OUT
cat > $OUT/run.py <<PY
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "visual-odometry-rs_amd"))
import numpy as np, torch, time
import vors_amd as V
rows, cols, L, n, F = 480, 640, 6, 64, 40
mode = {"c2f": 0, "dense": 1, "dso": 2}["$MODE"]
intr = V.scaled_intrinsics(rows, cols)
cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_FUSED)
# the sequences of bench.py sequences_bench
base = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])
rng = np.random.default_rng(11)
speed = 0.5 + 1.0 * rng.random(n)
sign = rng.choice([-1.0, 1.0], size=(n, 6))
blocky = (1 << 63) if mode == 2 else 0
frames = [V.synth_render_frames([blocky | (4242 + s) for s in range(n)], [k] * n, [base * sign[s] * speed[s] * k for s in range(n)], rows, cols, intr)
          for k in range(F)]
t = V.Trackers(cfg, n, rows, cols)
for _ in range(2):  # the first pass warms up
    t.init(*frames[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(1, F):
        t.track(*frames[k])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
poses, status, kf = t.current_frames()
print(f"$MODE: {n * (F - 1) / dt:.0f} frames/s, {dt / (F - 1) * 1e3:.3f} ms per lock-step frame, keyframes now at frame indices {sorted(set(kf.tolist()))[:8]}...")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o seq -- python $OUT/run.py > $OUT/run.log 2>&1
grep "frames/s" $OUT/run.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$OUT/kernel_stats.md", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats: 64 lock-step sequences x 39 tracked frames, 640x480, 6 levels, $MODE (tools/seq_profile.sh)\n\n")
    o.write([l for l in open("$OUT/run.log").read().splitlines() if "frames/s" in l][-1] + "\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:14]:
        o.write(f"| {r['Name'][:90]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |\n")
print(open("$OUT/kernel_stats.md").read()[:1500])
PY
find $OUT -name "*_kernel_trace.csv" -delete
