"""How much of a pose hangs on choices the reference does not pin (CPU only).

The LM loop branches on f32 comparisons (E_new > E_old, dE > 1: lm_optimizer.rs:144,179), so 1-ulp differences can change the
number of iterations. This test measures, on BASELINE-sized pairs (640x480, 6 levels), how far the POSE moves when

  * the 29 sums are accumulated in f64 instead of the reference's sequential f32 (`acc64`: the summation order is the one thing
    the GPU's tree reduction cannot reproduce), and
  * each "nalgebra assumption" of the oracle (nalgebra 0.17 is not vendored in the reference) is swapped for the other plausible
    evaluation order (`nalg1` quaternion product, `nalg2` dot folds, `nalg4` q * v, `nalg8` Cholesky / solve, `nalg16`
    from_quaternion).

Every variant flips accept/reject branches in a large share of the pairs and none moves a pose by more than a few 1e-6 — two
orders of magnitude below the 1e-4 bar. That is the evidence that the bar is safe although iteration counts are not reproducible.
"""
import numpy as np
import pytest

from oracle import oracle as O

VARIANTS = ("acc64", "nalg1", "nalg2", "nalg4", "nalg8", "nalg16")
BOUND = 2e-5  # observed: <= 4e-6


@pytest.mark.parametrize("mode,n", [(0, 48), (1, 6)], ids=["coarse_to_fine", "dense"])
def test_pose_sensitivity_to_unpinned_choices(mode, n):
    rows, cols, L = 480, 640, 6
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, _, _ = O.synth_batch(n, rows, cols, seed0=0x5EEDC000, intr=intr)
    cfg = O.make_config(L, intr, candidates_mode=mode)
    ref = O.track_pairs(cfg, kg, kd, cg, n_threads=8)
    assert (ref["status"] == 0).all()
    for v in VARIANTS:
        r = O.track_pairs(cfg, kg, kd, cg, n_threads=8, variant=v)
        d = np.abs(r["poses"] - ref["poses"]).max(axis=1)
        flips = (r["nb_iter"] != ref["nb_iter"]).any(axis=1).mean()
        print(f"mode {mode} {v}: max pose delta {d.max():.2e}, median {np.median(d):.2e}, branch-flip rate {flips:.0%}")
        assert (r["status"] == ref["status"]).all()
        assert (r["n_points"] == ref["n_points"]).all()
        assert d.max() < BOUND, f"{v} moves a pose by {d.max():.2e}"


def test_variants_really_differ():
    """Guard against a probe that silently compiles to the same arithmetic: over a batch each variant must change at least one bit."""
    rows, cols, L, n = 120, 160, 4, 24
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, _, _ = O.synth_batch(n, rows, cols, seed0=0x5EEDC100, intr=intr, motion_scale=2.0)
    cfg = O.make_config(L, intr)
    ref = O.track_pairs(cfg, kg, kd, cg)
    for v in VARIANTS:
        r = O.track_pairs(cfg, kg, kd, cg, variant=v)
        assert (r["poses"].view(np.uint32) != ref["poses"].view(np.uint32)).any(), v
        assert np.abs(r["poses"] - ref["poses"]).max() < 1e-4
