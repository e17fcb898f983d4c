"""Sequential tracking of ONE synthetic sequence (BASELINE configs 1/3 shape: Tracker::track per frame, keyframe switches
included) through the C ABI tracker, beside the CPU oracle on the same frames. Latency-bound by construction ("replicas only",
DESIGN.md §5): reports frames/s and the pose agreement. Usage: python tools/sequence_bench.py [n_frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first)
import vors_amd as V
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rows, cols, L = 480, 640, 6
intr = O.INTRINSICS_FR1
step = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])
frames_smooth = [O.synth_frame(31337, step * k, rows, cols, intr, frame_salt=k, n_threads=8) for k in range(n)]
frames_blocky = [O.synth_frame((1 << 63) | 31337, step * k, rows, cols, intr, frame_salt=k, n_threads=8) for k in range(n)]
for mode in (0, 1, 2):
    frames = frames_blocky if mode == 2 else frames_smooth  # the DSO selector needs mostly-flat images with edges
    cfg = V.Config(nb_levels=L, intrinsics=V.INTRINSICS_FR1, candidates_mode=mode, arithmetic=int(os.environ.get("ARITH", "1")))
    vt = cfg.init(0.0, frames[0][1], 0.0, frames[0][0])
    vt.track(0.0, frames[1][1], 0.0, frames[1][0])  # warm-up
    vt = cfg.init(0.0, frames[0][1], 0.0, frames[0][0])
    t0 = time.perf_counter(); sw = 0
    gp = []
    for k in range(1, n):
        vt.track(float(k), frames[k][1], float(k), frames[k][0])
        gp.append(vt.current_frame()[1]); sw += int(vt.last_stats()["change_keyframe"])
    tg = time.perf_counter() - t0
    ot = O.Tracker(O.make_config(L, intr, candidates_mode=mode), 0.0, frames[0][1], 0.0, frames[0][0], keep_debug=False)
    t0 = time.perf_counter()
    op = []
    for k in range(1, n):
        ot.track(float(k), frames[k][1], float(k), frames[k][0]); op.append(ot.current_frame()[1])
    tc = time.perf_counter() - t0
    err = np.abs(np.array(gp) - np.array(op)).max(axis=1)
    gt = O.iso_inverse(O.gt_model7(step * (n - 1)))
    print(f"mode {mode}: {n-1} frames, {sw} keyframe switches: GPU tracker {(n-1)/tg:.0f} frames/s ({tg/(n-1)*1e3:.2f} ms/frame), "
          f"CPU oracle {(n-1)/tc:.1f} frames/s; max pose diff {err.max():.2e}; final pose vs ground truth {np.abs(gp[-1]-gt).max():.2e}")
