"""How many pairs are still iterating after r evaluation rounds (dense mode)? Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
import numpy as np, torch
import vors_amd as V
rows, cols, L, n = 480, 640, 6, 4096
intr = V.scaled_intrinsics(rows, cols)
cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=1, arithmetic=V.ARITH_FUSED)
b = V.Batch(cfg, n, rows, cols)
kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000, n, rows, cols, intr)
poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda"); stats = V.stats_tensor(n)
b.track_pairs(kg, kd, cg, poses, status, stats); torch.cuda.synchronize()
st = V.decode_stats(stats)
it = st["nb_iter"][:, :L]; gr = st["nb_grad_evals"][:, :L]
print("mean iterations per level (0..5):", np.round(it.mean(0), 2), " mean grad evals:", np.round(gr.mean(0), 2))
# rounds per level in the split path: init (1) + per iteration: energy round (+1 g/H round when accepted and continuing)
rounds = np.zeros(n, int)
for l in (1, 0):
    cont = np.maximum(gr[:, l] - 2, 0) + (it[:, l] > 1) * 0   # accepted candidates beyond the first that went on (approx.)
    rounds += 1 + it[:, l] + np.maximum(gr[:, l] - 1 - 1, 0)
for r in (4, 5, 6, 8, 10, 12, 16, 20, 24, 30, 40):
    print(f"pairs needing more than {r} rounds: {(rounds > r).sum()}")
print("max rounds", rounds.max(), "hist of it[0]:", np.bincount(it[:, 0])[:24], "it[1]:", np.bincount(it[:, 1])[:24])
