"""Development aid (run through gpurun): frames/s of the lock-step sequence workload of bench.py sequences_64 alone, no profiler — for A/B of a
scheduling knob: VORS_LM_BLOCK=512 python tools/seq_time.py dense|c2f|dso"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "visual-odometry-rs_amd"))
import numpy as np, torch, time
import vors_amd as V
rows, cols, L, n, F = 480, 640, 6, 64, 40
mode = {"c2f": 0, "dense": 1, "dso": 2}[sys.argv[1]]
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
for _ in range(4):  # the first passes warm up
    t.init(*frames[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(1, F):
        t.track(*frames[k])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
poses, status, kf = t.current_frames()
print(f"{sys.argv[1]}: {n * (F - 1) / dt:.0f} frames/s, {dt / (F - 1) * 1e3:.3f} ms per lock-step frame, keyframes now at frame indices {sorted(set(kf.tolist()))[:8]}...")
