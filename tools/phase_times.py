"""Phases of the per-pair LM kernel at ONE level (needs a build with -DVORS_PROFILE_PHASES=<level>: tools/build_variant.sh ph5 -DVORS_PROFILE_PHASES=5):
shader-clock cycles per evaluation spent in the point loop / the workgroup reduction / the one-lane step / everything else.
Development aid; run through gpurun.   env: PAIRS (1), MODES ("0"), VLIB, LEVEL (must match the build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
V.LIB_PATH = os.path.join(ROOT, "visual-odometry-rs_amd", "vors_amd", os.environ.get("VLIB", "libvors_hip_eph5.so"))
rows, cols, L = 480, 640, 6
lvl = int(os.environ.get("LEVEL", "5"))
intr = V.scaled_intrinsics(rows, cols)
n = int(os.environ.get("PAIRS", "1"))
poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda"); stats = V.stats_tensor(n)
for mode in [int(x) for x in os.environ.get("MODES", "0").split(",")]:
    kg, kd, cg, _, gt = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
    b = V.Batch(V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=int(os.environ.get('ARITH', '1'))), n, rows, cols)
    for _ in range(3): b.track_pairs(kg, kd, cg, poses, status, stats)
    torch.cuda.synchronize()
    st = V.decode_stats(stats)
    ev = (st["nb_iter"][:, lvl] + 1).astype(float)
    loop, red, step, other = st["n_points"][:, 6], st["n_points"][:, 7], st["nb_iter"][:, 6], st["nb_iter"][:, 7]
    tot = loop + red + step + other
    print(f"mode {mode} level {lvl} ({st['n_points'][:, lvl].mean():.0f} points, {ev.mean():.1f} evaluations): cycles per evaluation: point loop {np.mean(loop / ev):.0f}, "
          f"reduction {np.mean(red / ev):.0f}, step {np.mean(step / (ev - 1).clip(1)):.0f}, other {np.mean(other / ev):.0f}; total {np.mean(tot / ev):.0f} "
          f"(= {np.mean(tot / ev) / 2400:.2f} us at 2.4 GHz)")
    del b
