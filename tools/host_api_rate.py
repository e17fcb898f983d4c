"""PCIe-inclusive rate of the host-buffer entry vors_track_pairs (allocation + H2D + compute + D2H per call). Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
from oracle import oracle as O
rows, cols, L, n = 480, 640, 6, 256
intr = O.scaled_intrinsics(rows, cols)
kg, kd, cg, _, gt = V.synth_render_pairs(0x5EED0000, n, rows, cols, intr)
kg, kd, cg = kg.cpu().numpy(), kd.cpu().numpy().view(np.uint16), cg.cpu().numpy()
for mode in (0, 1):
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode)
    V.track_pairs(cfg, kg, kd, cg)
    t0 = time.perf_counter()
    for _ in range(3): V.track_pairs(cfg, kg, kd, cg)
    dt = (time.perf_counter() - t0) / 3
    print(f"mode {mode}: vors_track_pairs (host buffers, pageable memory) {n} pairs in {dt*1e3:.1f} ms -> {n/dt:.0f} pairs/s ({n*4*rows*cols/dt/1e9:.1f} GB/s of input)")
