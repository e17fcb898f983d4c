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


IDEPTH_SHAPES = [(64, 96, 4, ""), (128, 160, 5, ""), (128, 160, 5, "1"), (60, 80, 3, ""), (36, 64, 4, ""), (50, 72, 3, ""), (51, 77, 3, "")]


@pytest.mark.parametrize("rows,cols,L,level12", IDEPTH_SHAPES, ids=[f"{c}x{r}_L{l}{'_level12' if e else ''}" for r, c, l, e in IDEPTH_SHAPES])
def test_dense_inverse_depth_pyramid_bit_exact_on_every_path(rows, cols, L, level12, monkeypatch):
    """Dense mode's inverse-depth pyramid (inverse_depth.rs:24-29,81-98) has four device forms by shape — levels 1-3 in one pass
    (rows % 8 == 0, cols % 8 == 0, >= 4 levels), levels 1-2 in one pass (rows % 4 == 0, cols % 16 == 0; also behind
    VORS_IDEPTH_LEVEL12=1), level 1 with wide loads (cols % 8 == 0) and the per-pixel form — and one halving kernel for the rest. Every
    level of every form == the oracle tracker's points: coordinates, inverse depths and Jacobians bit for bit, with a third of the
    depths unknown so that every child count occurs."""
    import torch
    if level12:
        monkeypatch.setenv("VORS_IDEPTH_LEVEL12", level12)
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = O.synth_batch(2, rows, cols, seed0=0x5EEDAB00 + rows, intr=intr)
    rng = np.random.default_rng(rows + cols)
    kd[rng.random(kd.shape) < 0.33] = 0
    kd[1, : rows // 2, : cols // 2] = 0   # whole blocks unknown at every level
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=1)
    b = V.Batch(cfg, 2, rows, cols)
    poses = torch.zeros((2, 7), device="cuda"); status = torch.zeros(2, dtype=torch.int32, device="cuda")
    b.track_pairs(torch.from_numpy(kg).cuda(), torch.from_numpy(kd.view(np.int16)).cuda(), torch.from_numpy(cg).cuda(), poses, status)
    torch.cuda.synchronize()
    ocfg = O.make_config(L, intr, candidates_mode=1)
    for p in range(2):
        tr = O.Tracker(ocfg, 0.0, kd[p], 0.0, kg[p])
        for l in range(L):
            xy, iz, jac, tm = b.points(p, l)
            oxy, oiz, ojac = tr.points(l)
            assert xy.shape == oxy.shape, f"pair {p} level {l}: {len(xy)} vs {len(oxy)} points"
            o1, o2 = np.lexsort((xy[:, 1], xy[:, 0])), np.lexsort((oxy[:, 1], oxy[:, 0]))
            assert (xy[o1] == oxy[o2]).all()
            assert (np.ascontiguousarray(iz[o1]).view(np.uint32) == np.ascontiguousarray(oiz[o2]).view(np.uint32)).all(), f"pair {p} level {l}"
            assert (np.ascontiguousarray(jac[o1]).view(np.uint32) == np.ascontiguousarray(ojac[o2]).view(np.uint32)).all(), f"pair {p} level {l}"
