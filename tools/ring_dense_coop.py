"""Development aid (gpurun): dense REFERENCE, 512 pairs per step through a ring of 3 / 6 handles, by workgroup size of the workgroup-per-pair kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L, n = 480, 640, 6, 512
intr = V.scaled_intrinsics(rows, cols)
cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=1, arithmetic=V.ARITH_REFERENCE)
sets = []
for k in range(6):
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 + k * n, n, rows, cols, intr)
    sets.append((kg, kd, cg, torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")))
for coop in ("", "0", "3", "4", "5", "8"):
    if coop: os.environ["VORS_REF_COOP"] = coop
    row = []
    for depth in (1, 3, 6):
        pipe = V.Pipeline(cfg, n, rows, cols, depth=depth)
        def step(i):
            s = sets[i % depth]; pipe.submit(s[0], s[1], s[2], s[3], s[4])
        for i in range(depth): step(i)
        pipe.drain(); torch.cuda.synchronize()
        K = 12
        t0 = time.perf_counter()
        for i in range(K): step(i)
        pipe.drain(); torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / K * 1e3)
        del pipe
    print(f"dense REFERENCE 512 pairs, VORS_REF_COOP={coop or 'default'}: ring 1 {row[0]:.3f} | ring 3 {row[1]:.3f} | ring 6 {row[2]:.3f} ms per step", flush=True)
