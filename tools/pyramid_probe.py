"""Development aid (run through gpurun): time of the pyramid stage (both pyramids of a step) by number of levels, fused launch vs one level
per launch (VORS_PYRAMID_FUSED=0), at 4096 and at 1 pair(s).   usage: python tools/pyramid_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))

if len(sys.argv) > 1:
    import numpy as np, torch
    import vors_amd as V
    rows, cols = 480, 640
    intr = V.scaled_intrinsics(rows, cols)
    for n in (4096, 1):
        kg = torch.randint(0, 256, (n, rows, cols), dtype=torch.uint8, device="cuda")
        cg = torch.randint(0, 256, (n, rows, cols), dtype=torch.uint8, device="cuda")
        kd = torch.zeros((n, rows, cols), dtype=torch.int16, device="cuda")   # no depth: no candidates, the LM stage is empty
        poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda")
        line = []
        for L in (2, 3, 4, 6):
            cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]))
            b = V.Batch(cfg, n, rows, cols)
            b.enable_kernel_timing(32)
            for _ in range(24):
                b.track_pairs(kg, kd, cg, poses, status)
            torch.cuda.synchronize()
            line.append(f"L{L}: {(np.median(b.kernel_times('pyramid_keyframe')[-16:]) + np.median(b.kernel_times('pyramid_current')[-16:])) * 1e3:.1f} us")
            del b
        print(f"[{sys.argv[1]}] {n} pair(s): " + ", ".join(line), flush=True)
else:
    for tag, env in (("fused", {}), ("per level", {"VORS_PYRAMID_FUSED": "0"})):
        subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, **env))
