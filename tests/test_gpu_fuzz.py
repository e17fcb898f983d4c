"""Seeded random shapes through the C ABI against the oracle: image sizes with every residue mod 4 / 8 / 16 (they select different
load widths and fallbacks in the kernels), all pyramid depths the size allows, the three candidate modes, optional Huber weights,
skewed / negative intrinsics. Small images so the oracle finishes in seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vors_amd as V
from oracle import oracle as O

POSE_TOL = 1e-4   # rad / metres (BASELINE.json north_star)
BLOCKY = 1 << 63  # seeds with the top bit set render the piecewise-constant texture (synth_scene.h) the DSO selector needs


def _cases():
    rng = np.random.default_rng(0xF022)
    out = []
    for i in range(36):
        rows, cols = int(rng.integers(24, 150)), int(rng.integers(24, 200))
        lmax = 1
        while min(rows, cols) >> lmax >= 2 and lmax < 8:
            lmax += 1
        L = int(rng.integers(1, lmax + 1))
        mode = int(rng.integers(0, 3))
        huber = float(rng.choice([0.0, 0.0, 8.0]))
        out.append((i, rows, cols, L, mode, huber))
    return out


@pytest.mark.parametrize("arith", [V.ARITH_EXACT, V.ARITH_FUSED], ids=["exact", "fused"])
@pytest.mark.parametrize("i,rows,cols,L,mode,huber", _cases(), ids=lambda v: str(v))
def test_random_shapes_vs_oracle(i, rows, cols, L, mode, huber, arith):
    import torch
    rng = np.random.default_rng(1000 + i)
    intr = list(O.scaled_intrinsics(rows, cols))
    if i % 5 == 0:
        intr[4] = 0.2                       # skew
    if i % 7 == 0:
        intr[3] = -intr[3]                  # negative fv like ICL-NUIM (tum_rgbd.rs:25)
    seed0 = (0x5EEDF000 + 16 * i) | (BLOCKY if mode == 2 else 0)
    n = 3
    kg, kd, cg, cd, gt = O.synth_batch(n, rows, cols, seed0=seed0, intr=tuple(intr), motion_scale=float(rng.uniform(0.3, 1.5)))
    ref = O.track_pairs(O.make_config(L, tuple(intr), candidates_mode=mode, huber_delta=huber), kg, kd, cg)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(tuple(intr[:2]), tuple(intr[2:4]), intr[4]), candidates_mode=mode, huber_delta=huber,
                   arithmetic=arith)
    b = V.Batch(cfg, n, rows, cols)
    t = [torch.from_numpy(np.ascontiguousarray(kg)).cuda(), torch.from_numpy(np.ascontiguousarray(kd).view(np.int16)).cuda(),
         torch.from_numpy(np.ascontiguousarray(cg)).cuda()]
    poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
    status = torch.zeros(n, dtype=torch.int32, device="cuda")
    stats = V.stats_tensor(n)
    b.track_pairs(*t, poses, status, stats)
    torch.cuda.synchronize()
    st = V.decode_stats(stats)
    assert (status.cpu().numpy() == ref["status"]).all()
    assert (st["n_points"][:, :L] == ref["n_points"]).all()
    ok = ref["status"] == 0
    p = poses.cpu().numpy()
    err = np.abs(p - ref["poses"]).max(axis=1)
    same_path = (st["nb_iter"][:, :L] == ref["nb_iter"]).all(axis=1)
    assert (err[ok] < POSE_TOL).all(), f"pose error {err} (rows={rows} cols={cols} L={L} mode={mode})"
    assert (p[~ok] == ref["poses"][~ok]).all()
    if (ok & ~same_path).any():
        # an accept / reject comparison within rounding of a tie: iteration counts differ, the poses still agree
        print(f"[fuzz {i}] {int((ok & ~same_path).sum())} pair(s) took another accept/reject path; pose diff {err[ok & ~same_path]}")
