"""Generates tests/golden/*.npz: small synthetic frame pairs with the ORACLE's outputs for every stage.

The reference cannot run in this image (pure Rust, no toolchain), so these vectors pin the oracle restatement (against
regressions) and give the GPU tests fixed expected values that travel to the GPU box. Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, rows, cols, L, n, mode, seed0, thresh=7, motion_scale=1.0):
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = O.synth_batch(n, rows, cols, seed0=seed0, intr=intr, motion_scale=motion_scale)
    cfg = O.make_config(L, intr, thresh=thresh, candidates_mode=mode)
    out = dict(rows=rows, cols=cols, L=L, mode=mode, thresh=thresh, intr=np.asarray(intr, np.float64), kf_gray=kg, kf_depth=kd,
               cur_gray=cg, cur_depth=cd, gt=gt)
    res = O.track_pairs(cfg, kg, kd, cg, cd)
    out.update(poses=res["poses"], models=res["models"], status=res["status"], nb_iter=res["nb_iter"], n_points=res["n_points"],
               flow=res["flow"])
    tr = O.Tracker(cfg, 0.0, kd[0], 0.0, kg[0])
    out["mask0"] = tr.mask()
    cur_pyr = O.mean_pyramid(cg[0], L)
    for l in range(L):
        xy, iz, jac = tr.points(l)
        out[f"img{l}"] = tr.image(l)
        out[f"xy{l}"], out[f"iz{l}"], out[f"jac{l}"] = xy, iz, jac
        _, _, _, k = tr.level(l)
        out[f"k{l}"] = k
        # operator-level evaluation at the identity and at the final model
        for tag, model in (("id", np.array([0, 0, 0, 0, 0, 0, 1], np.float32)), ("fin", res["models"][0])):
            e, ni, g, H, r = O.lm_eval(k, tr.image(l), cur_pyr[l], xy, iz, jac, model, want_residuals=True)
            out[f"ev_{tag}{l}_e"], out[f"ev_{tag}{l}_n"], out[f"ev_{tag}{l}_g"], out[f"ev_{tag}{l}_H"], out[f"ev_{tag}{l}_r"] = \
                np.float32(e), np.int32(ni), g, H, r
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written:", {l: len(out[f"iz{l}"]) for l in range(L)}, "nb_iter[0]", res["nb_iter"][0])


if __name__ == "__main__":
    make("sparse_128x96_L4", 96, 128, 4, 6, 0, 0x5EED1000)
    make("sparse_odd_167x123_L3", 123, 167, 3, 3, 0, 0x5EED2000)
    make("dense_80x60_L3", 60, 80, 3, 3, 1, 0x5EED3000)
