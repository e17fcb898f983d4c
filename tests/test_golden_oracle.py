"""The oracle must reproduce the committed golden vectors bit for bit (pins the restatement against regressions)."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    L = int(g["L"])
    cfg = O.make_config(L, tuple(g["intr"]), thresh=int(g["thresh"]), candidates_mode=int(g["mode"]))
    res = O.track_pairs(cfg, g["kf_gray"], g["kf_depth"], g["cur_gray"], g["cur_depth"])
    assert (res["poses"].view(np.uint32) == g["poses"].view(np.uint32)).all()
    assert (res["nb_iter"] == g["nb_iter"]).all() and (res["status"] == g["status"]).all()
    tr = O.Tracker(cfg, 0.0, g["kf_depth"][0], 0.0, g["kf_gray"][0])
    assert (tr.mask() == g["mask0"]).all()
    for l in range(L):
        xy, iz, jac = tr.points(l)
        assert (xy == g[f"xy{l}"]).all() and (iz == g[f"iz{l}"]).all() and (jac == g[f"jac{l}"]).all()
        assert (tr.image(l) == g[f"img{l}"]).all()


def test_synthetic_generator_is_deterministic():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sparse_128x96_L4.npz"))
    kg, kd, cg, cd, gt = O.synth_pair(0x5EED1000, int(g["rows"]), int(g["cols"]), tuple(g["intr"]))
    assert (kg == g["kf_gray"][0]).all() and (kd == g["kf_depth"][0]).all() and (cg == g["cur_gray"][0]).all()
