"""Development aid (gpurun): ring depth sweep of vors_pipeline_* at 512 pairs per step (BASELINE config 4's per-GPU share) and 4096."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
for aname, arith in (("fused", V.ARITH_FUSED), ("reference", V.ARITH_REFERENCE)):
    for mode, mname in ((0, "c2f"), (2, "dso")):
        for n in (512, 4096):
            cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)
            depths = (1, 2, 3, 4, 6)
            sets = []
            for k in range(max(depths)):
                kg, kd, cg, _, _ = V.synth_render_pairs((0x5EED0000 + k * n) | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
                sets.append((kg, kd, cg, torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")))
            row = []
            for depth in depths:
                pipe = V.Pipeline(cfg, n, rows, cols, depth=depth)
                def step(i):
                    s = sets[i % depth]; pipe.submit(s[0], s[1], s[2], s[3], s[4])
                for i in range(2 * depth): step(i)
                pipe.drain(); torch.cuda.synchronize()
                K = 100 if n == 512 else 40
                t0 = time.perf_counter()
                for i in range(K): step(i)
                pipe.drain(); torch.cuda.synchronize()
                row.append((time.perf_counter() - t0) / K * 1e3)
                del pipe
            print(f"{aname:9s} {mname:4s} {n:5d} pairs: " + " | ".join(f"ring {d}: {t:.3f} ms" for d, t in zip(depths, row)), flush=True)
