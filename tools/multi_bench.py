"""Throughput of the torch-free multi-GPU entry (vors_multi_track_pairs: ONE process, all visible devices, one RCCL all-gather per step).
bench.py --gpus N measures the same partitioning with one process per GPU through torch.distributed (the driver's contract); this is the
C-ABI form a Rust host would use.   usage: python tools/multi_bench.py [pairs_per_gpu] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V

per = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows, cols, L = 480, 640, 6
nd = V.device_count()
intr = V.scaled_intrinsics(rows, cols)
cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=V.CANDIDATES_DENSE, arithmetic=V.ARITH_FUSED)
m = V.MultiGpu(cfg, per, rows, cols)
n = per * nd
shards = [[], [], []]
for k in range(nd):
    with torch.cuda.device(k):
        kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 + k * per, per, rows, cols, intr, device=f"cuda:{k}")
        for lst, t in zip(shards, (kg, kd, cg)):
            lst.append(t)
for k in range(nd):
    torch.cuda.synchronize(k)
for _ in range(2):
    m.track_pairs(*shards, n)
t0 = time.perf_counter()
for _ in range(steps):
    poses, status = m.track_pairs(*shards, n)
dt = time.perf_counter() - t0
print(f"{nd} device(s) x {per} pairs: {n * steps / dt:.0f} frame-pairs/s ({dt / steps * 1e3:.2f} ms per step, gather and D2H included); failed pairs {int((status != 0).sum())}")
