"""Throughput of both candidate modes at several LM block sizes (development aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
if os.environ.get("VLIB"): V.LIB_PATH = V.LIB_PATH.replace("libvors_hip.so", os.environ["VLIB"])
from oracle import oracle as O

rows, cols, L = 480, 640, 6
intr = O.scaled_intrinsics(rows, cols)
npairs = int(os.environ.get("PAIRS", "256"))
kg, kd, cg, _, gt = V.synth_render_pairs(0x5EED0000, npairs, rows, cols, intr)
poses = torch.zeros((npairs, 7), dtype=torch.float32, device="cuda"); status = torch.zeros(npairs, dtype=torch.int32, device="cuda")
stats = V.stats_tensor(npairs)
ref = {}
for mode in (0, 1):
    for blk in [int(x) for x in os.environ.get("BLOCKS", "256,512,1024").split(",")]:
        os.environ["VORS_LM_BLOCK"] = str(blk)
        cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode)
        b = V.Batch(cfg, npairs, rows, cols); b.enable_kernel_timing(16)
        for _ in range(2): b.track_pairs(kg, kd, cg, poses, status, stats)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): b.track_pairs(kg, kd, cg, poses, status, stats)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        p = poses.cpu().numpy()
        if mode not in ref: ref[mode] = p
        print(f"mode {mode} block {blk}: {dt*1e3:.3f} ms/step {npairs/dt:.0f} pairs/s  lm {b.kernel_times('lm')[-10:].mean():.3f} kf {b.kernel_times('keyframe')[-10:].mean():.3f} "
              f"pyr {b.kernel_times('pyramid_keyframe')[-10:].mean()+b.kernel_times('pyramid_current')[-10:].mean():.3f} ms  maxdiff vs first {np.abs(p-ref[mode]).max():.2e} failed {int(status.sum())}")
        del b
