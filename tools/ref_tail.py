"""Development aid (gpurun): how uneven is the work of the pairs of a batch in the REFERENCE arithmetic (groups of 64 points evaluated per pair =
sum over levels of (nb_iter + 1) * ceil(n_points / 64)), and what the LM stage would take without the imbalance (a batch of identical pairs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L = 480, 640, 6
intr = V.scaled_intrinsics(rows, cols)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, mode in (("c2f", 0), ("dso", 2)):
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_REFERENCE)
    def run(kg, kd, cg):
        poses, status = torch.zeros((n, 7), device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
        stats = V.stats_tensor(n)
        b = V.Batch(cfg, n, rows, cols)
        b.enable_kernel_timing(16)
        for _ in range(6):
            b.track_pairs(kg, kd, cg, poses, status, stats)
        torch.cuda.synchronize()
        return float(b.kernel_times("lm")[-4:].mean()), V.decode_stats(stats)
    lm, st = run(kg, kd, cg)
    work = ((st["nb_iter"][:, :L] + 1) * ((st["n_points"][:, :L] + 63) // 64)).sum(axis=1)
    q = np.percentile(work, [50, 90, 99, 100])
    print(f"{name} {n} pairs: lm {lm:.3f} ms | groups per pair: mean {work.mean():.0f} median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} max {q[3]:.0f} (max / mean {q[3] / work.mean():.2f})")
    # the same batch size made of copies of the pair closest to the mean work
    i = int(np.argmin(np.abs(work - work.mean())))
    rep = lambda t: t[i:i + 1].expand(n, *t.shape[1:]).contiguous()
    lm2, st2 = run(rep(kg), rep(kd), rep(cg))
    w2 = ((st2["nb_iter"][:, :L] + 1) * ((st2["n_points"][:, :L] + 63) // 64)).sum(axis=1)
    print(f"   {n} copies of pair {i} ({w2.mean():.0f} groups each): lm {lm2:.3f} ms -> {lm2 / w2.mean() * work.mean():.3f} ms at the batch's mean work")
