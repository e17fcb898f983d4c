"""Per-frame latency distribution of the single-sequence tracker (Tracker::track through the C ABI), per candidate mode and arithmetic.
Development aid; run through gpurun."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
from oracle import oracle as O
n=120; rows, cols, L = 480, 640, 6
intr = O.INTRINSICS_FR1
step = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])
frames_smooth = [O.synth_frame(31337, step * k, rows, cols, intr, frame_salt=k, n_threads=8) for k in range(n)]
frames_blocky = [O.synth_frame((1 << 63) | 31337, step * k, rows, cols, intr, frame_salt=k, n_threads=8) for k in range(n)]
for rep in range(3):
  for mode in (0, 1, 2):
    frames = frames_blocky if mode == 2 else frames_smooth
    for arith, aname in ((V.ARITH_FUSED, "fused"), (V.ARITH_REFERENCE, "reference")):
        cfg = V.Config(nb_levels=L, intrinsics=V.INTRINSICS_FR1, candidates_mode=mode, arithmetic=arith)
        vt = cfg.init(0.0, frames[0][1], 0.0, frames[0][0])
        vt.track(0.0, frames[1][1], 0.0, frames[1][0])
        vt = cfg.init(0.0, frames[0][1], 0.0, frames[0][0])
        ts=[]
        for k in range(1, n):
            t0 = time.perf_counter(); vt.track(float(k), frames[k][1], float(k), frames[k][0]); ts.append(time.perf_counter()-t0)
        ts=np.array(ts)*1e3
        print(f"rep {rep} mode {mode} arith {aname}: mean {ts.mean():.3f} ms median {np.median(ts):.3f} p90 {np.quantile(ts,0.9):.3f} max {ts.max():.3f}")
