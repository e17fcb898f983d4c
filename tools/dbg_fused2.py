import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'visual-odometry-rs_amd')
import numpy as np, torch
import vors_amd as V
from oracle import oracle as O
rows, cols, L, n = 120, 160, 4, 6
intr = O.scaled_intrinsics(rows, cols)
kg, kd, cg, _, gt = O.synth_batch(n, rows, cols, seed0=0x5EEDF500 + rows, intr=intr)
cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=1)
b = V.Batch(cfg, n, rows, cols)
t = [torch.from_numpy(kg).cuda(), torch.from_numpy(kd.view(np.int16)).cuda(), torch.from_numpy(cg).cuda()]
poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda")
b.track_pairs(*t, poses, status); torch.cuda.synchronize()
np.set_printoptions(linewidth=220, precision=7)
for pair in (0, 4):
    for lvl in range(L):
        for model in (np.array([0, 0, 0, 0, 0, 0, 1], np.float32), gt[pair]):
            e0, n0, g0, H0 = b.eval_level(pair, lvl, model, 0)
            e1, n1, g1, H1 = b.eval_level(pair, lvl, model, 1)
            print(f"pair {pair} lvl {lvl}: n {n0} {n1}  E {e0:.6f} {e1:.6f} rel {abs(e1-e0)/e0:.2e}  g rel {np.abs(g1-g0).max()/np.abs(g0).max():.2e}  H rel {np.abs(H1-H0).max()/np.abs(H0).max():.2e}")
