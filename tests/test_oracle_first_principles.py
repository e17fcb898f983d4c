"""The oracle against FIRST PRINCIPLES (CPU only, float64 numpy / scipy) — not against itself, and not against another reading of
the Rust text.

The reference holds no vectors for the tracker's arithmetic ("parity unpinned"), so the oracle's formulas are checked
here against what they must mean mathematically:

  * se3::exp / se3::log (se3.rs:65-129)            vs the matrix exponential / logarithm of the 4x4 twist matrix (scipy.linalg)
  * warp_jacobian_at (inverse_compositional.rs:313-341) vs central differences of  pi(K, expm(hat(xi)) . pi^-1(x, y, 1/_z))  — pins the
    twist ordering (3 linear, then 3 angular), every sign, the skew terms and the use of the gradient at (y, x)
  * eval_energy + compute_eval_data (lm_optimizer.rs:68-107) vs a vectorised float64 evaluation written from the camera model and the
    definition of bilinear interpolation (back-project, rigid transform, project, strict window, residual, sum of J r and J J^T)
  * step (lm_optimizer.rs:123-136)                 vs numpy.linalg.solve on the damped normal equations + expm for the update
  * the inverse-depth pyramid (inverse_depth.rs:49-98 with strategy_dso_mean) vs its closed form: the plain mean of the level-0
    candidates' inverse depths inside the 2^l block (all weights are equal at level 0, and weights add)

Tolerances are float32 rounding of the quantities involved; they are written next to each assertion.
"""
import numpy as np
import pytest
from scipy.linalg import expm, logm

from oracle import oracle as O

ROWS, COLS, LEVELS = 240, 320, 5


def hat(xi):
    """4x4 twist matrix of xi = (v, w): [[w]x v; 0 0] (se3.rs:33-41: linear part first)."""
    v, w = xi[:3], xi[3:]
    return np.array([[0, -w[2], w[1], v[0]], [w[2], 0, -w[0], v[1]], [-w[1], w[0], 0, v[2]], [0, 0, 0, 0]], np.float64)


def quat_to_rot(q):
    i, j, k, w = (float(x) for x in q)
    return np.array([[1 - 2 * (j * j + k * k), 2 * (i * j - k * w), 2 * (i * k + j * w)],
                     [2 * (i * j + k * w), 1 - 2 * (i * i + k * k), 2 * (j * k - i * w)],
                     [2 * (i * k - j * w), 2 * (j * k + i * w), 1 - 2 * (i * i + j * j)]], np.float64)


def iso_to_mat(m7):
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(m7[3:])
    T[:3, 3] = np.asarray(m7[:3], np.float64)
    return T


def back_project(intr, x, y, z):
    cu, cv, fu, fv, s = (float(a) for a in intr)
    Y = (y - cv) * z / fv
    X = ((x - cu) * z - s * Y) / fu
    return np.stack([X, Y, z * np.ones_like(X)], axis=-1)


def project(intr, P):
    cu, cv, fu, fv, s = (float(a) for a in intr)
    return (fu * P[..., 0] + s * P[..., 1] + cu * P[..., 2]) / P[..., 2], (fv * P[..., 1] + cv * P[..., 2]) / P[..., 2]


@pytest.fixture(scope="module")
def scene():
    intr = O.scaled_intrinsics(ROWS, COLS)
    kg, kd, cg, cd, gt = O.synth_pair(0x5EEDF001, ROWS, COLS, intr, motion_scale=1.0)
    tr = O.Tracker(O.make_config(LEVELS, intr), 0.0, kd, 0.0, kg)
    return dict(intr=intr, kg=kg, kd=kd, cg=cg, gt=gt, tr=tr, cur_pyr=O.mean_pyramid(cg, LEVELS))


def test_se3_exp_and_log_are_the_matrix_exponential_and_logarithm():
    rng = np.random.default_rng(1)
    for scale in (1e-3, 5e-2, 0.7, 2.5):
        for _ in range(25):
            xi = (rng.uniform(-1, 1, 6) * scale).astype(np.float32)
            T = expm(hat(xi.astype(np.float64)))
            m = O.se3_exp(xi)
            assert np.allclose(iso_to_mat(m), T, atol=3e-6 * max(1.0, scale))  # f32 rounding of a rotation / a translation of size ~scale
            if np.linalg.norm(xi[3:]) < 3.0:  # the logarithm is unique below pi
                back = O.se3_log(m)
                L = logm(iso_to_mat(m)).real
                ref = np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])
                assert np.allclose(back, ref, atol=2e-4 * max(1.0, scale))  # se3.rs:142 tests its own round trip at 1e-4


@pytest.mark.parametrize("lvl", [0, 2, 4])
def test_warp_jacobian_is_the_derivative_of_the_warp(scene, lvl):
    tr = scene["tr"]
    _, _, n, k = tr.level(lvl)
    xy, z, jac = tr.points(lvl)
    gx, gy, _ = tr.gradients(lvl)
    assert n > 50
    sel = np.linspace(0, n - 1, 200).astype(int)
    x, y = xy[sel, 0].astype(np.float64), xy[sel, 1].astype(np.float64)
    P = back_project(k, x, y, 1.0 / z[sel].astype(np.float64))
    Ph = np.concatenate([P, np.ones((len(sel), 1))], axis=1)
    gu = gx[xy[sel, 1], xy[sel, 0]].astype(np.float64)  # gradient looked up at (row = y, col = x)
    gv = gy[xy[sel, 1], xy[sel, 0]].astype(np.float64)
    eps = 1e-6
    J = np.zeros((len(sel), 6))
    for q in range(6):
        e = np.zeros(6)
        e[q] = eps
        up, vp = project(k, (Ph @ expm(hat(e)).T)[:, :3])
        um, vm = project(k, (Ph @ expm(hat(-e)).T)[:, :3])
        J[:, q] = gu * (up - um) / (2 * eps) + gv * (vp - vm) / (2 * eps)
    scale = np.abs(J).max(axis=1, keepdims=True) + 1e-9
    # f32 evaluation of ~10 operations vs f64 central differences: observed 1.5e-7; swapping the linear and angular halves gives 1.4,
    # a flipped sign 2.0
    assert (np.abs(jac[sel] - J) / scale).max() < 5e-6


@pytest.mark.parametrize("lvl,perturb", [(0, 0.0), (1, 0.002), (3, 0.01)])
def test_evaluation_sums_against_float64_definition(scene, lvl, perturb):
    tr = scene["tr"]
    rows, cols, n, k = tr.level(lvl)
    xy, z, jac = tr.points(lvl)
    tmpl = tr.image(lvl)
    img = scene["cur_pyr"][lvl]
    rng = np.random.default_rng(lvl)
    model = O.iso_mul(scene["gt"], O.se3_exp((rng.uniform(-1, 1, 6) * perturb).astype(np.float32)))
    energy, n_in, g, H, res = O.lm_eval(k, tmpl, img, xy, z, jac, model, want_residuals=True)

    T = iso_to_mat(model)
    P = back_project(k, xy[:, 0].astype(np.float64), xy[:, 1].astype(np.float64), 1.0 / z.astype(np.float64))
    u, v = project(k, P @ T[:3, :3].T + T[:3, 3])
    fu, fv = np.floor(u), np.floor(v)
    inside = (fu >= 0) & (fu < cols - 2) & (fv >= 0) & (fv < rows - 2)  # lm_optimizer.rs:227-231: strict, excludes the last TWO columns / rows
    # points within float32 rounding of a window border or of an integer coordinate may fall either way: leave them out of the comparison
    safe = inside & (np.minimum(u - fu, fu + 1 - u) > 1e-3) & (np.minimum(v - fv, fv + 1 - v) > 1e-3)
    iu, iv = fu[safe].astype(int), fv[safe].astype(int)
    a, b = (u - fu)[safe], (v - fv)[safe]
    I = img.astype(np.float64)
    val = (1 - a) * (1 - b) * I[iv, iu] + a * (1 - b) * I[iv, iu + 1] + (1 - a) * b * I[iv + 1, iu] + a * b * I[iv + 1, iu + 1]
    r = val - tmpl[xy[safe, 1], xy[safe, 0]].astype(np.float64)
    # (deterministic inputs, IEEE f32 without contraction in the oracle: the few borderline points fall the same way on every host)
    assert int(inside.sum()) == n_in
    assert np.abs(res[safe] - r).max() < 2e-3  # grey levels (0..255) through ~25 f32 operations at coordinates of a few hundred pixels; observed 6e-4
    # the sums, over the same point set; the handful of points left out above enter with the oracle's own residuals
    unsafe = inside & ~safe
    Jd = jac.astype(np.float64)
    r_all = np.zeros(n)
    r_all[safe] = r
    r_all[unsafe] = res[unsafe]
    g_ref = (Jd[inside] * r_all[inside, None]).sum(0)
    H_ref = Jd[inside].T @ Jd[inside]
    e_ref = (r_all[inside] ** 2).mean()
    assert np.abs(g - g_ref).max() <= 5e-4 * np.abs(Jd[inside] * r_all[inside, None]).sum(0).max()  # observed 7e-5 (sequential f32 sums)
    assert np.abs(H - H_ref).max() <= 1e-4 * np.abs(H_ref).max()                                      # observed 1e-6
    assert abs(energy - e_ref) <= 1e-4 * e_ref                                                        # observed 2e-7 .. 2e-5


def test_step_solves_the_damped_normal_equations():
    rng = np.random.default_rng(7)
    for trial in range(40):
        J = rng.normal(size=(60, 6)) * rng.uniform(0.1, 30, 6)
        H = (J.T @ J).astype(np.float32)
        g = (J.T @ rng.normal(size=60)).astype(np.float32)
        lam = float(rng.choice([1e-3, 0.1, 10.0]))
        model = O.se3_exp((rng.uniform(-0.2, 0.2, 6)).astype(np.float32))
        st, new, delta = O.lm_step(H, g, model, lam)
        assert st == 0
        Hd = H.astype(np.float64).copy()
        Hd[np.diag_indices(6)] *= 1.0 + lam  # lm_optimizer.rs:126-129
        d_ref = np.linalg.solve(Hd, g.astype(np.float64))
        assert np.allclose(delta, d_ref, rtol=2e-3 * np.linalg.cond(Hd) ** 0.5 / 10 + 1e-3, atol=1e-5 * np.abs(d_ref).max())
        T_ref = iso_to_mat(model) @ np.linalg.inv(expm(hat(delta.astype(np.float64))))  # model * exp(delta)^-1, lm_optimizer.rs:135
        assert np.allclose(iso_to_mat(new), T_ref, atol=5e-6 * max(1.0, np.abs(T_ref[:3, 3]).max()))
    # a matrix that is not positive definite: cholesky() is None -> Err (lm_optimizer.rs:131-133)
    H = np.eye(6, dtype=np.float32)
    H[3, 3] = -1.0
    st, _, _ = O.lm_step(H, np.ones(6, np.float32), np.array([0, 0, 0, 0, 0, 0, 1], np.float32), 0.1)
    assert st != 0


@pytest.mark.parametrize("lvl", [1, 2, 4])
def test_inverse_depth_pyramid_is_the_block_mean_of_the_candidates(scene, lvl):
    tr = scene["tr"]
    mask = tr.mask().astype(bool)
    depth = scene["kd"].astype(np.float64)
    known = mask & (depth > 0)
    iz0 = np.where(known, 5000.0 / np.where(depth > 0, depth, 1.0), 0.0)  # from_depth (inverse_depth.rs:24-29), depth_scale 5000
    b = 1 << lvl
    r, c = (ROWS // b) * b, (COLS // b) * b
    cnt = known[:r, :c].reshape(r // b, b, c // b, b).sum(axis=(1, 3))
    tot = iz0[:r, :c].reshape(r // b, b, c // b, b).sum(axis=(1, 3))
    xy, z, _ = tr.points(lvl)
    rows_l, cols_l, n, _ = tr.level(lvl)
    assert (rows_l, cols_l) == (ROWS >> lvl, COLS >> lvl)
    assert n == int((cnt > 0).sum())  # a level-l point exists iff its block holds a candidate with a depth
    assert (cnt[xy[:, 1], xy[:, 0]] > 0).all()
    ref = tot[xy[:, 1], xy[:, 0]] / cnt[xy[:, 1], xy[:, 0]]
    assert np.abs(z - ref).max() <= 4e-6 * np.abs(ref).max() * (lvl + 1)  # a few f32 roundings per fused level
    # column-major enumeration (x outer, y inner): extract_z, inverse_compositional.rs:266-277
    order = np.lexsort((xy[:, 1], xy[:, 0]))
    assert (order == np.arange(n)).all()
