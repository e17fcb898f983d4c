"""The oracle against every known answer the reference holds for this path, CPU only.

Reference tests / KATs covered (paths relative to the reference repository):
  src/core/candidates/coarse_to_fine.rs:68-71   prune_with_thresh examples
  src/math/so3.rs:114-143                      exp_log_round_trip, hat_vee_roundtrip, hat_2_ok, log_exp_round_trip
  src/math/se3.rs:144-173                      exp_log_round_trip, hat_vee_roundtrip, log_exp_round_trip
  src/dataset/tum_rgbd.rs:15-52                constants
"""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import oracle as O

f32 = st.floats(min_value=-100.0, max_value=100.0, width=32, allow_nan=False, allow_infinity=False)


def test_prune_with_thresh_doc_examples():
    assert O.prune_with_thresh(5, 0, 1, 8, 9) == [False, False, True, True]
    assert O.prune_with_thresh(5, 0, 9, 1, 8) == [False, True, False, True]
    assert O.prune_with_thresh(5, 1, 0, 9, 0) == [False, False, True, False]


def test_prune_ties_and_wrap():
    # flat block: stable ascending sort keeps d last -> d alone
    assert O.prune_with_thresh(7, 3, 3, 3, 3) == [False, False, False, True]
    # u16 wrapping add of `third + thresh` (release build semantics): 65535 + 7 wraps to 6 -> second (65535) > 6
    assert O.prune_with_thresh(7, 65535, 65535, 65535, 0) == [False, True, True, False]


def test_so3_se3_zero_roundtrip_exact():
    assert (O.so3_log(O.so3_exp(np.zeros(3))) == 0).all()
    assert (O.se3_log(O.se3_exp(np.zeros(6))) == 0).all()


@settings(max_examples=200, deadline=None)
@given(f32, f32, f32)
def test_so3_hat_vee_and_hat2(x, y, z):
    w = np.array([x, y, z], np.float32)
    assert (O.so3_vee(O.so3_hat(w)) == w).all()
    h = O.so3_hat(w).reshape(3, 3)
    # hat_2(x) == hat(x) * hat(x) exactly (each entry is a sum of two products in both forms)
    prod = np.zeros((3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            acc = np.float32(0)
            for k in range(3):
                acc = np.float32(acc + np.float32(h[i, k] * h[k, j]))
            prod[i, j] = acc
    assert (O.so3_hat_2(w).reshape(3, 3) == prod).all()


@settings(max_examples=200, deadline=None)
@given(f32, f32, f32, f32, f32, f32)
def test_se3_hat_vee(v1, v2, v3, w1, w2, w3):
    xi = np.array([v1, v2, v3, w1, w2, w3], np.float32)
    assert (O.se3_vee(O.se3_hat(xi)) == xi).all()


def _quat_from_euler(roll, pitch, yaw):
    """nalgebra UnitQuaternion::from_euler_angles (the reference tests' generator, so3.rs:146-148)."""
    sr, cr = np.sin(roll * 0.5), np.cos(roll * 0.5)
    sp, cp = np.sin(pitch * 0.5), np.cos(pitch * 0.5)
    sy, cy = np.sin(yaw * 0.5), np.cos(yaw * 0.5)
    w = cr * cp * cy + sr * sp * sy
    i = sr * cp * cy - cr * sp * sy
    j = cr * sp * cy + sr * cp * sy
    k = cr * cp * sy - sr * sp * cy
    return np.array([i, j, k, w], np.float32)


def _rel_eq_up_to_sign(a, b, eps):
    def rel(a, b):
        return np.all(np.abs(a - b) <= np.maximum(eps, eps * np.maximum(np.abs(a), np.abs(b))))
    return rel(a, b) or rel(a, -b)  # nalgebra's quaternion RelativeEq accounts for the double cover


@settings(max_examples=300, deadline=None)
@given(f32, f32, f32)
def test_so3_log_exp_round_trip(roll, pitch, yaw):
    q = _quat_from_euler(roll, pitch, yaw)
    q2 = O.so3_exp(O.so3_log(q))
    assert _rel_eq_up_to_sign(q, q2, 2e-6)  # reference tolerance 1e-6 (so3.rs:112); 2e-6 absorbs generator rounding


@settings(max_examples=300, deadline=None)
@given(f32, f32, f32, f32, f32, f32)
def test_se3_log_exp_round_trip(t1, t2, t3, a1, a2, a3):
    q = _quat_from_euler(a1, a2, a3)
    iso = np.array([t1, t2, t3, *q], np.float32)
    iso2 = O.se3_exp(O.se3_log(iso))
    eps = 1e-4  # se3.rs:142
    assert _rel_eq_up_to_sign(iso[3:], iso2[3:], eps)
    assert np.all(np.abs(iso[:3] - iso2[:3]) <= np.maximum(eps, 2 * eps * np.maximum(np.abs(iso[:3]).max(), 1.0)))


def test_se3_exp_closed_form():
    """exp alone against an independent float64 Rodrigues formula (the reference never tests this)."""
    rng = np.random.default_rng(0)
    for _ in range(200):
        xi = rng.uniform(-1, 1, 6) * rng.choice([1e-3, 1e-2, 0.3, 2.0])
        out = O.se3_exp(xi.astype(np.float32))
        v, w = xi[:3].astype(np.float32).astype(np.float64), xi[3:].astype(np.float32).astype(np.float64)
        th = np.linalg.norm(w)
        W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        Vm = np.eye(3) + (1 - np.cos(th)) / th**2 * W + (th - np.sin(th)) / th**3 * W @ W
        q = np.concatenate([np.sin(th / 2) / th * w, [np.cos(th / 2)]])
        assert np.allclose(out[:3], Vm @ v, atol=2e-6, rtol=1e-5)
        assert np.allclose(out[3:], q, atol=2e-6)


def test_mean_pyramid_matches_numpy_and_stops_early():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 50), dtype=np.uint8)
    pyr = O.mean_pyramid(img, 10)
    # 37x50 -> 18x25 -> 9x12 -> 4x6 -> 2x3 -> 1x1 ; then a side < 2 stops the sequence (multires.rs:73-77)
    assert [p.shape for p in pyr] == [(37, 50), (18, 25), (9, 12), (4, 6), (2, 3), (1, 1)]
    for a, b in zip(pyr[:-1], pyr[1:]):
        r, c = b.shape
        a = a[:2 * r, :2 * c].astype(np.uint16)
        exp = ((a[0::2, 0::2] + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2]) // 4).astype(np.uint8)
        assert (exp == b).all()
    assert len(O.mean_pyramid(img, 0)) == 1 and len(O.mean_pyramid(img, 1)) == 1


def test_intrinsics_multires_keeps_skew():
    k = O.intrinsics_multires([318.64304, 255.313989, 517.306408, 516.469215, 0.25], 3)
    assert np.allclose(k[1], [(318.64304 + 0.5) / 2 - 0.5, (255.313989 + 0.5) / 2 - 0.5, 517.306408 / 2, 516.469215 / 2, 0.25])
    assert k[2][4] == np.float32(0.25)  # camera.rs:121: skew is NOT scaled


def test_gradients_and_candidates_integer_semantics():
    rng = np.random.default_rng(2)
    rows, cols, L = 48, 64, 3
    img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    depth = rng.integers(1, 60000, (rows, cols), dtype=np.uint16)
    tr = O.Tracker(O.make_config(L, O.scaled_intrinsics(rows, cols)), 0.0, depth, 0.0, img)
    gx, gy, g2 = tr.gradients(0)
    i = img.astype(np.int32)
    ex = np.zeros_like(i); ey = np.zeros_like(i)
    ex[1:-1, 1:-1] = np.trunc((i[1:-1, 2:] - i[1:-1, :-2]) / 2)  # truncation toward zero (gradient.rs:28-29)
    ey[1:-1, 1:-1] = np.trunc((i[2:, 1:-1] - i[:-2, 1:-1]) / 2)
    assert (gx == ex).all() and (gy == ey).all()
    assert (g2 == ((ex * ex + ey * ey) & 0xFFFF)).all()
    # block gradients of level 1 come from level 0 (multires.rs:112-126); squared norm wraps mod 65536
    gx1, gy1, g21 = tr.gradients(1)
    a, b, c, d = i[0::2, 0::2], i[1::2, 0::2], i[0::2, 1::2], i[1::2, 1::2]
    assert (gx1 == np.trunc((c + d - a - b) / 2)).all() and (gy1 == np.trunc((b - a + d - c) / 2)).all()
    assert (g21 == ((gx1.astype(np.int32) ** 2 + gy1.astype(np.int32) ** 2) & 0xFFFF)).all()
    # mask structure: every coarsest pixel keeps 1 or 2 of 4 children at each step
    mask = tr.mask().astype(bool)
    n0 = mask.sum()
    n_roots = (rows >> (L - 1)) * (cols >> (L - 1))
    assert n_roots * 1 <= n0 <= n_roots * 2 ** (L - 1)
    blocks = mask.reshape(rows // 2, 2, cols // 2, 2).sum(axis=(1, 3))
    assert set(np.unique(blocks)) <= {0, 1, 2}


def test_oracle_tracker_recovers_ground_truth():
    kg, kd, cg, cd, gt = O.synth_pair(0x5EED0003, 240, 320)
    tr = O.Tracker(O.make_config(5, O.scaled_intrinsics(240, 320)), 0.0, kd, 0.0, kg)
    assert tr.track(1.0, cd, 1.0, cg) == 0
    last = tr.last()
    assert np.abs(last["lm_model"] - gt).max() < 5e-3
    t, pose = tr.current_frame()
    assert t == 1.0
    assert np.allclose(O.iso_mul(pose, last["lm_model"]), [0, 0, 0, 0, 0, 0, 1], atol=1e-6)  # pose = model^-1 (keyframe at identity)


def test_oracle_failure_semantics_pose_kept():
    """No usable candidate (all depths unknown): energy NaN, Cholesky fails, pose is kept."""
    kg, kd, cg, cd, _ = O.synth_pair(0x5EED0004, 64, 96)
    tr = O.Tracker(O.make_config(3, O.scaled_intrinsics(64, 96)), 0.0, np.zeros_like(kd), 0.0, kg)
    assert tr.track(1.0, cd, 1.0, cg) == 1
    _, pose = tr.current_frame()
    assert (pose == np.array([0, 0, 0, 0, 0, 0, 1], np.float32)).all()
    assert tr.last()["went_well"] is False


def test_oracle_pyramid_too_short_is_an_error():
    kg, kd, _, _, _ = O.synth_pair(1, 16, 16)
    with pytest.raises(ValueError):
        O.Tracker(O.make_config(6, O.scaled_intrinsics(16, 16)), 0.0, kd, 0.0, kg)


def test_dso_selector_structure():
    """dso::select (examples/candidates_dso.rs parameters) on a piecewise-constant scene: about the target number of points,
    every pick is a block maximum, recursion / sub-sampling branches are reachable, and the result is repeatable."""
    B = 1 << 63
    kg, kd, cg, cd, gt = O.synth_pair(B | 11, 480, 640)
    m, rounds = O.dso_mask(kg)
    assert 1600 <= m.sum() <= 2400 and rounds == [4]
    m2, _ = O.dso_mask(kg)
    assert (m == m2).all()
    # at most one pick per 4x4 block (base size 4, level 1) and none on the 1-px border (gradient magnitude is 0 there)
    assert m.reshape(120, 4, 160, 4).sum(axis=(1, 3)).max() == 1
    assert m[0].sum() == 0 and m[-1].sum() == 0 and m[:, 0].sum() == 0 and m[:, -1].sum() == 0
    small = O.synth_pair(B | 12, 120, 160, O.scaled_intrinsics(120, 160))[0]
    assert O.dso_mask(small)[1][:2] == [4, 2]                      # too few candidates -> smaller blocks, second round
    # the smooth texture has large gradient MAGNITUDES everywhere: squared-median thresholds reject everything (as the reference would)
    assert O.dso_mask(O.synth_pair(13, 240, 320)[0])[0].sum() == 0
