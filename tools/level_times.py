"""Per-level time of the per-pair LM kernel (needs a VORS_PROFILE_LEVELS build: tools/build_variant.sh prof -DVORS_PROFILE_LEVELS=1).
Development aid; run through gpurun.   env: PAIRS (4096), MODES ("0,2"), ARITH (1), VLIB (libvors_hip_eprof.so)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
V.LIB_PATH = os.path.join(ROOT, "visual-odometry-rs_amd", "vors_amd", os.environ.get("VLIB", "libvors_hip_eprof.so"))
rows, cols, L = 480, 640, int(os.environ.get("LEVELS", "6"))
intr = V.scaled_intrinsics(rows, cols)
n = int(os.environ.get("PAIRS", "4096"))
poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda"); stats = V.stats_tensor(n)
for mode in [int(x) for x in os.environ.get("MODES", "0,2").split(",")]:
    kg, kd, cg, _, gt = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
    b = V.Batch(V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=int(os.environ.get('ARITH', '1'))), n, rows, cols)
    b.enable_kernel_timing(4)
    for _ in range(3): b.track_pairs(kg, kd, cg, poses, status, stats)
    torch.cuda.synchronize()
    st = V.decode_stats(stats)
    if L <= 6: print(f"   epilogue mean {st['energy'][:, 6].mean():.1f} us, whole workgroup mean {st['energy'][:, 7].mean():.1f} max {st['energy'][:, 7].max():.1f} us")
    us = st["energy"][:, :L]
    it = st["nb_iter"][:, :L]
    print(f"mode {mode}: lm kernel {b.kernel_times('lm')[-1]*1e3:.0f} us; points per level {np.round(st['n_points'][:, :L].mean(0))}; per-level mean us {np.round(us.mean(0),1)} max {np.round(us.max(0),1)}; "
          f"mean evals {np.round((it+1).mean(0),1)}; sum-of-levels mean {us.sum(1).mean():.0f} max {us.sum(1).max():.0f}; us/eval {np.round(us.mean(0)/(it+1).mean(0),2)}")
    del b
