"""multires::mean_pyramid on the device (one fused launch for up to five halvings, `pyramid_fused_kernel`; one level per launch for widths
that are not multiples of 16, for user buffers that are not 16-byte aligned and for levels beyond the sixth) == the oracle's mean_pyramid
(multires.rs:21-31,67-88), every level, bit for bit, at sizes that exercise: partial tiles, odd sizes at every level (halve floors), widths
that take the per-level path, 6 levels (all five fused halvings), 7 and 8 levels (fused + more launches), unaligned user buffers."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
from oracle import oracle as O
import vors_amd as V

pytestmark = pytest.mark.gpu

SHAPES = [(480, 640, 6), (960, 1280, 7), (1024, 1280, 8), (64, 128, 7), (65, 144, 6), (127, 112, 5), (250, 336, 6), (33, 48, 5),
          (96, 100, 4), (101, 203, 5), (480, 640, 1), (480, 640, 2), (66, 272, 3)]


@pytest.mark.parametrize("rows,cols,L", SHAPES, ids=[f"{c}x{r}_L{l}" for r, c, l in SHAPES])
def test_every_pyramid_level_equals_the_oracle(rows, cols, L):
    import torch
    rng = np.random.default_rng(rows * 10007 + cols)
    n = 3
    kg = rng.integers(0, 256, (n, rows, cols), dtype=np.uint8)
    cg = rng.integers(0, 256, (n, rows, cols), dtype=np.uint8)
    kg[1, : rows // 2] = 255   # saturated area: (255 * 4) >> 2 must not wrap
    kd = rng.integers(0, 30000, (n, rows, cols)).astype(np.uint16)
    intr = O.scaled_intrinsics(rows, cols)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]))
    b = V.Batch(cfg, n, rows, cols)
    poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda")
    for unaligned in (False, True):
        if unaligned:
            raw_k = torch.zeros(n * rows * cols + 5, dtype=torch.uint8, device="cuda")
            raw_c = torch.zeros(n * rows * cols + 5, dtype=torch.uint8, device="cuda")
            t_kg, t_cg = raw_k[5:].view(n, rows, cols), raw_c[5:].view(n, rows, cols)
            t_kg.copy_(torch.from_numpy(kg)); t_cg.copy_(torch.from_numpy(cg))
        else:
            t_kg, t_cg = torch.from_numpy(kg).cuda(), torch.from_numpy(cg).cuda()
        b.track_pairs(t_kg, torch.from_numpy(kd.view(np.int16)).cuda(), t_cg, poses, status)
        torch.cuda.synchronize()
        for p in range(n):
            want_k, want_c = O.mean_pyramid(kg[p], L), O.mean_pyramid(cg[p], L)
            assert len(want_k) == L
            for l in range(L):
                assert (b.keyframe_image(p, l) == want_k[l]).all(), f"keyframe pair {p} level {l} unaligned={unaligned}"
                assert (b.current_image(p, l) == want_c[l]).all(), f"current pair {p} level {l} unaligned={unaligned}"
