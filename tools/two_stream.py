"""Experiment: alternate steps between two batch handles on two streams (each step = one full pass over its 4096-pair batch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L = 480, 640, 6
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = {"dense": 1, "c2f": 0}[sys.argv[2] if len(sys.argv) > 2 else "dense"]
intr = V.scaled_intrinsics(rows, cols)
cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_FUSED)
for nstreams in (1, 2, 3):
    ws = []
    for k in range(nstreams):
        b = V.Batch(cfg, n, rows, cols)
        kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 + k * n, n, rows, cols, intr)
        ws.append((b, kg, kd, cg, torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"), torch.cuda.Stream()))
    torch.cuda.synchronize()
    def step(i):
        b, kg, kd, cg, p, s, st = ws[i % nstreams]
        with torch.cuda.stream(st):
            b.track_pairs(kg, kd, cg, p, s)
    for i in range(6): step(i)
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nstreams} stream(s): {n * K / dt:.0f} pairs/s, {dt / K * 1e3:.3f} ms per step")
    del ws
