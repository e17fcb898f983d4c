"""VORS_ARITH_REFERENCE: the HIP path with the reference's own summation order must equal the CPU oracle EXACTLY — not within a tolerance.

What is asserted here, through the C ABI, is equality of BITS: candidate lists in extract_z's column-major order
(inverse_compositional.rs:260-279), the 29 sums of an evaluation (lm_optimizer.rs:68-107), every iteration count of every level
(optimizer.rs:57-70), the optical flow of the keyframe test (inverse_compositional.rs:213-224), the final model and the pose — for
pairs, for sequences with keyframe switches, in the three candidate modes, with and without the Huber extension, at the operator level
and for the lock-step trackers. EXACT and FUSED are then gated against THIS mode (tests/test_gpu_fused.py). GPU only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vors_amd as V
from oracle import oracle as O

BLOCKY = 1 << 63
MODES = {0: "coarse_to_fine", 1: "dense", 2: "dso"}


def vcfg(L, intr, mode=0, thresh=7, huber=0.0):
    return V.Config(nb_levels=L, candidates_diff_threshold=thresh, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]),
                    candidates_mode=mode, huber_delta=huber, arithmetic=V.ARITH_REFERENCE)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    return (bits(a) == bits(b)).all()


def run_batch(cfg, kg, kd, cg, prev=None):
    import torch
    n, rows, cols = kg.shape
    b = V.Batch(cfg, n, rows, cols)
    t = (torch.from_numpy(np.ascontiguousarray(kg)).cuda(), torch.from_numpy(np.ascontiguousarray(kd).view(np.int16)).cuda(),
         torch.from_numpy(np.ascontiguousarray(cg)).cuda())
    poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
    status = torch.zeros(n, dtype=torch.int32, device="cuda")
    stats = V.stats_tensor(n)
    if prev is not None:
        prev_t = torch.from_numpy(np.ascontiguousarray(prev, np.float32)).cuda()
        b.track_pairs(*t, poses, status, stats, prev_poses7=prev_t)
    else:
        b.track_pairs(*t, poses, status, stats)
    torch.cuda.synchronize()
    return b, poses.cpu().numpy(), status.cpu().numpy(), V.decode_stats(stats), t


def synth(n, rows, cols, intr, seed0, blocky=False):
    return O.synth_batch(n, rows, cols, seed0=(BLOCKY if blocky else 0) | seed0, intr=intr)


def assert_pairs_identical(ref, poses, status, stats, L, what):
    assert (status == ref["status"]).all(), what
    assert (stats["n_points"][:, :L] == ref["n_points"]).all(), what
    assert (stats["nb_iter"][:, :L] == ref["nb_iter"]).all(), \
        f"{what}: iteration counts differ in pairs {np.nonzero((stats['nb_iter'][:, :L] != ref['nb_iter']).any(axis=1))[0][:8]}"
    assert same_bits(stats["lm_model"], ref["models"]), f"{what}: final models differ by {np.abs(stats['lm_model'] - ref['models']).max():.3e}"
    assert same_bits(poses, ref["poses"]), f"{what}: poses differ by {np.abs(poses - ref['poses']).max():.3e}"
    assert same_bits(stats["optical_flow"], ref["flow"]), what


# ---------------------------------------------------------------------------------------------- lists in the reference's order
@pytest.mark.parametrize("mode", [0, 2], ids=["coarse_to_fine", "dso"])
@pytest.mark.parametrize("rows,cols,L", [(120, 160, 4), (240, 320, 5), (123, 167, 3), (480, 640, 6)])
def test_candidate_lists_come_out_in_extract_z_order(mode, rows, cols, L):
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = synth(2, rows, cols, intr, 0x5EED7100 + rows, blocky=(mode == 2))
    b, *_ = run_batch(vcfg(L, intr, mode), kg, kd, cg)
    for p in range(2):
        tr = O.Tracker(O.make_config(L, intr, candidates_mode=mode), 0.0, kd[p], 0.0, kg[p])
        for l in range(L):
            xy_o, iz_o, jac_o = tr.points(l)
            xy, iz, jac, tm = b.points(p, l)
            assert len(iz) == len(iz_o)
            assert (xy == xy_o).all(), f"pair {p} level {l}: the list is not in column-major order"
            assert same_bits(iz, iz_o) and same_bits(jac, jac_o)
            # column-major means: sorted by x, then y, strictly
            key = xy[:, 0].astype(np.int64) * (rows >> l) + xy[:, 1]
            assert (np.diff(key) > 0).all()


@pytest.mark.parametrize("mode", [0, 2], ids=["coarse_to_fine", "dso"])
def test_both_forms_of_the_column_major_sort_give_the_same_lists(monkeypatch, mode):
    """Round 4: sort_colmajor_kernel keeps lists of at most 4096 records in registers and writes them back in place; longer lists take the
    multi-pass form through the scratch copy, which VORS_REF_SORT_REGCAP=0 forces for every list (1000: level 0 only). Same lists, hence
    the same poses bit for bit — and both equal the oracle's."""
    rows, cols, L, n = 480, 640, 6, 6
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = synth(n, rows, cols, intr, 0x5EED7700, blocky=(mode == 2))
    ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg, kd, cg)
    res = {}
    # (round 5: the coarse-to-fine mode ranks the keyframe kernel's staged regions directly — rank_regions_kernel, compaction and sort in one
    # pass; VORS_REF_RANK=0 brings back compact_regions_kernel + sort_colmajor_kernel, whose two forms are then exercised as before)
    monkeypatch.setenv("VORS_REF_RANK", "0")
    for cap in ("", "0", "1000"):
        if cap:
            monkeypatch.setenv("VORS_REF_SORT_REGCAP", cap)
        else:
            monkeypatch.delenv("VORS_REF_SORT_REGCAP", raising=False)
        b, poses, status, stats, _ = run_batch(vcfg(L, intr, mode), kg, kd, cg)
        assert_pairs_identical(ref, poses, status, stats, L, f"sort form {cap or 'default'}")
        res[cap] = [[b.points(p, l)[0] for l in range(L)] for p in range(n)]
    monkeypatch.delenv("VORS_REF_SORT_REGCAP", raising=False)
    monkeypatch.delenv("VORS_REF_RANK", raising=False)
    b, poses, status, stats, _ = run_batch(vcfg(L, intr, mode), kg, kd, cg)
    assert_pairs_identical(ref, poses, status, stats, L, "default (coarse-to-fine: ranked straight from the staged regions)")
    res["rank"] = [[b.points(p, l)[0] for l in range(L)] for p in range(n)]
    for cap in ("0", "1000", "rank"):
        for p in range(n):
            for l in range(L):
                assert (res[cap][p][l] == res[""][p][l]).all(), (cap, p, l)


# ---------------------------------------------------------------------------------------------- operator level
@pytest.mark.parametrize("huber", [0.0, 10.0], ids=["l2", "huber10"])
def test_lm_eval_and_solve_on_explicit_observations_equal_the_oracle_bit_for_bit(huber):
    rows, cols, L = 120, 160, 4
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = O.synth_pair(0x5EED5000, rows, cols, intr)
    tr = O.Tracker(O.make_config(L, intr), 0.0, kd, 0.0, kg)
    cur = O.mean_pyramid(cg, L)
    model = np.array([0, 0, 0, 0, 0, 0, 1], np.float32)
    rng = np.random.default_rng(3)
    for l in range(L - 1, -1, -1):
        xy, iz, jac = tr.points(l)
        _, _, _, k = tr.level(l)
        obs = V.Obs(k, tr.image(l), cur[l], xy, iz, jac, huber_delta=huber, arithmetic=V.ARITH_REFERENCE)
        for m in (model, O.gt_model7(rng.normal(0, 4e-3, 6))):
            e, n, g, H, res = V.lm_eval(obs, m, want_residuals=True)
            eo, no, go, Ho, reso = O.lm_eval(k, tr.image(l), cur[l], xy, iz, jac, m, huber_delta=huber, want_residuals=True)
            assert n == no
            inside = ~np.isnan(res)
            assert inside.sum() == n and (inside == ~np.isnan(reso)).all() and same_bits(res[inside], reso[inside])
            assert same_bits(e, eo), f"level {l}: energy {e!r} vs {eo!r}"
            assert same_bits(g, go), f"level {l}: gradient differs by {np.abs(g - go).max():.3e}"
            assert same_bits(H, Ho), f"level {l}: Hessian differs by {np.abs(H - Ho).max():.3e}"
        st, m_dev, it_dev, e_dev, lam_dev = V.lm_solve(obs, model)
        ost, m_or, it_or, e_or, lam_or = O.lm_solve(k, tr.image(l), cur[l], xy, iz, jac, model, huber_delta=huber)
        assert (st, it_dev) == (ost, it_or), f"level {l}: status / iterations {(st, it_dev)} vs {(ost, it_or)}"
        assert same_bits(m_dev, m_or) and same_bits(e_dev, e_or) and same_bits(lam_dev, lam_or)
        model = m_or


def test_the_sums_depend_on_the_order_of_the_observations():
    """The point of the mode: reversing the candidate list changes the last bits of the REFERENCE sums exactly as it changes the oracle's."""
    rows, cols, L = 120, 160, 4
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = O.synth_pair(0x5EED5001, rows, cols, intr)
    tr = O.Tracker(O.make_config(L, intr), 0.0, kd, 0.0, kg)
    cur = O.mean_pyramid(cg, L)
    xy, iz, jac = tr.points(0)
    _, _, _, k = tr.level(0)
    m = O.gt_model7(np.array([2e-3, -1e-3, 1e-3, 5e-4, -4e-4, 3e-4]))
    differs = False
    for order in (slice(None), slice(None, None, -1)):
        obs = V.Obs(k, tr.image(0), cur[0], xy[order], iz[order], jac[order], arithmetic=V.ARITH_REFERENCE)
        e, n, g, H = V.lm_eval(obs, m)
        eo, no, go, Ho = O.lm_eval(k, tr.image(0), cur[0], xy[order], iz[order], jac[order], m)
        assert n == no and same_bits(e, eo) and same_bits(g, go) and same_bits(H, Ho)
        if order != slice(None):
            differs = not (same_bits(g, g0) and same_bits(H, H0))
        g0, H0 = g, H
    assert differs, "the two orders were expected to differ in the last bits (they do in the oracle)"


# ---------------------------------------------------------------------------------------------- pairs
@pytest.mark.parametrize("rows,cols,L,n,mode", [(120, 160, 4, 24, 0), (240, 320, 5, 8, 0), (480, 640, 6, 8, 0), (97, 131, 3, 6, 0),
                                                 (64, 64, 1, 2, 0), (384, 512, 8, 3, 0), (200, 328, 7, 3, 0),
                                                 (120, 160, 4, 6, 1), (101, 135, 3, 4, 1), (66, 130, 2, 3, 1), (240, 320, 5, 3, 1),
                                                 # pyramids whose top level is ONE row / one column (round 6: the column-major records' multiply-high
                                                 # divisor has no value for rows == 1; such handles keep the gathering source)
                                                 (32, 64, 6, 3, 1), (64, 32, 6, 3, 1), (34, 70, 6, 2, 0),
                                                 (120, 160, 4, 8, 2), (240, 320, 5, 6, 2), (480, 640, 6, 4, 2)])
def test_track_pairs_equal_the_oracle_bit_for_bit(rows, cols, L, n, mode):
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = synth(n, rows, cols, intr, 0x5EED4000 + rows, blocky=(mode == 2))
    ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg, kd, cg, n_threads=8)
    b, poses, status, stats, _ = run_batch(vcfg(L, intr, mode), kg, kd, cg)
    assert_pairs_identical(ref, poses, status, stats, L, f"{cols}x{rows} L{L} {MODES[mode]}")


@pytest.mark.parametrize("mode", [0, 1, 2], ids=list(MODES.values()))
def test_huber_extension_equals_the_oracle_bit_for_bit(mode):
    rows, cols, L, n = 120, 160, 4, 6
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = synth(n, rows, cols, intr, 0x5EED4800, blocky=(mode == 2))
    ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode, huber_delta=10.0), kg, kd, cg, n_threads=8)
    b, poses, status, stats, _ = run_batch(vcfg(L, intr, mode, huber=10.0), kg, kd, cg)
    assert_pairs_identical(ref, poses, status, stats, L, f"huber {MODES[mode]}")


def test_initial_guess_and_large_motion_take_the_non_taylor_branch_of_se3_exp():
    """Steps of more than 0.01 rad leave the Taylor branch of se3::exp (se3.rs:71-87): sinf / cosf then decide the last bits of the
    candidate model — the restated glibc algorithm on the device (lie.h ref_sinf) against the platform libm in the oracle."""
    rows, cols, L, n = 240, 320, 5, 12
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = O.synth_batch(n, rows, cols, seed0=0x5EED4A00, intr=intr, motion_scale=4.0)
    prev = np.zeros((n, 7), np.float32)
    prev[:, 6] = 1.0
    prev[::2] = O.gt_model7(np.array([3e-3, 2e-3, -1e-3, 2e-3, -1e-3, 1e-3]))   # every other pair starts from a non-identity pose
    ref = O.track_pairs(O.make_config(L, intr), kg, kd, cg, init_poses7=prev, n_threads=8)
    b, poses, status, stats, _ = run_batch(vcfg(L, intr, 0), kg, kd, cg, prev=prev)
    assert_pairs_identical(ref, poses, status, stats, L, "large motion")
    assert (ref["nb_iter"].sum(axis=1) > 10).all()


def test_degenerate_pairs_fail_like_the_oracle():
    """No usable candidate (all depths unknown): Cholesky of a zero Hessian fails at the coarsest level, the pose is kept, status 1
    (inverse_compositional.rs:195-199,206-208); 0 / 0 optical flow = NaN."""
    rows, cols, L = 96, 128, 3
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, cd, gt = O.synth_batch(3, rows, cols, seed0=0x5EED4B00, intr=intr)
    kd[1] = 0
    for mode in (0, 1):
        ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg, kd, cg)
        b, poses, status, stats, _ = run_batch(vcfg(L, intr, mode), kg, kd, cg)
        assert status[1] == 1 == ref["status"][1]
        assert_pairs_identical(ref, poses, status, stats, L, f"degenerate {MODES[mode]}")


def test_full_size_batch_is_identical_and_deterministic():
    rows, cols, L, n = 480, 640, 6, 96
    intr = O.INTRINSICS_FR1
    kg, kd, cg, cd, gt = O.synth_batch(n, rows, cols, seed0=0x5EED0000, intr=intr)
    ref = O.track_pairs(O.make_config(L, intr), kg, kd, cg, n_threads=16)
    b, poses, status, stats, t = run_batch(vcfg(L, intr, 0), kg, kd, cg)
    assert_pairs_identical(ref, poses, status, stats, L, "640x480 coarse-to-fine")
    b2, poses2, *_ = run_batch(vcfg(L, intr, 0), kg[::-1].copy(), kd[::-1].copy(), cg[::-1].copy())
    assert same_bits(poses2[::-1], poses)   # independent of the position in the batch


# ---------------------------------------------------------------------------------------------- sequences
def make_sequences(n_seq, n_frames, rows, cols, intr, blocky):
    import torch
    base = np.array([0.012, -0.006, 0.004, 0.002, -0.003, 0.001])
    rng = np.random.default_rng(7)
    speed = 0.35 + 1.3 * rng.random(n_seq)
    sign = rng.choice([-1.0, 1.0], size=(n_seq, 6))
    frames = []
    for k in range(n_frames):
        seeds = [(BLOCKY if blocky else 0) | (1000 + s) for s in range(n_seq)]
        g, d = V.synth_render_frames(seeds, [k] * n_seq, [base * sign[s] * speed[s] * k for s in range(n_seq)], rows, cols, intr)
        frames.append((g, d))
    torch.cuda.synchronize()
    return frames


@pytest.mark.parametrize("mode", [0, 1, 2], ids=list(MODES.values()))
def test_lock_step_sequences_with_keyframe_switches_equal_the_oracle_tracker_bit_for_bit(mode):
    rows, cols, L, n_seq, n_frames = 120, 160, 4, 12, 12
    intr = O.scaled_intrinsics(rows, cols)
    frames = make_sequences(n_seq, n_frames, rows, cols, intr, blocky=(mode == 2))
    gray = np.stack([g.cpu().numpy() for g, d in frames])
    depth = np.stack([d.cpu().numpy().view(np.uint16) for g, d in frames])
    ref = O.track_sequences(O.make_config(L, intr, candidates_mode=mode), gray, depth, n_threads=8)
    many = V.Trackers(vcfg(L, intr, mode), n_seq, rows, cols)
    many.init(*frames[0])
    for k in range(1, n_frames):
        many.track(*frames[k])
        poses, status, kf_index = many.current_frames()
        st = many.stats()
        assert (status == ref["status"][:, k - 1]).all()
        assert (st["change_keyframe"] == ref["changed_keyframe"][:, k - 1]).all(), f"frame {k}: keyframe decisions differ"
        assert same_bits(poses, ref["poses"][:, k - 1]), f"frame {k}: poses differ by {np.abs(poses - ref['poses'][:, k - 1]).max():.3e}"
    assert ref["changed_keyframe"].sum() >= 6, "the trajectories were meant to switch keyframes"


@pytest.mark.parametrize("mode", [0, 2], ids=["coarse_to_fine", "dso"])
def test_single_tracker_equals_the_oracle_tracker_bit_for_bit(mode):
    rows, cols, L = 240, 320, 5
    intr = O.scaled_intrinsics(rows, cols)
    step = np.array([0.012, -0.006, 0.004, 0.002, -0.003, 0.001])
    seed = (BLOCKY | 77) if mode == 2 else 77
    frames = [O.synth_frame(seed, step * k, rows, cols, intr, frame_salt=k) for k in range(14)]
    ot = O.Tracker(O.make_config(L, intr, candidates_mode=mode), 0.0, frames[0][1], 0.0, frames[0][0])
    vt = vcfg(L, intr, mode).init(0.0, frames[0][1], 0.0, frames[0][0])
    switches = 0
    for k in range(1, len(frames)):
        g, d = frames[k]
        assert ot.track(0.1 * k, d, 0.1 * k + 0.01, g) == vt.track(0.1 * k, d, 0.1 * k + 0.01, g)
        (to, po), (tv, pv) = ot.current_frame(), vt.current_frame()
        assert to == tv and same_bits(po, pv), f"frame {k}: {np.abs(po - pv).max():.3e}"
        ol, vl = ot.last(), vt.last_stats()
        assert ol["changed_keyframe"] == bool(vl["change_keyframe"])
        assert (np.asarray(ol["nb_iter"])[:L] == np.asarray(vl["nb_iter"])[:L]).all()
        assert same_bits(ot.keyframe_pose()[1], vt.keyframe()[1])
        switches += int(ol["changed_keyframe"])
    assert switches >= 1


# ---------------------------------------------------------------------------------------------- the kernels behind the mode
@pytest.mark.parametrize("mode", [0, 1, 2], ids=["coarse_to_fine", "dense", "dso"])
def test_every_kernel_form_gives_the_oracles_bits(monkeypatch, mode):
    """Round 5: the REFERENCE arithmetic runs as one wavefront per pair (large batches), as a workgroup per pair (small batches: 2, 4 or 8
    wavefronts, one of them owning the chains) and as the first followed by the second for the pairs still iterating when most of the batch
    is done (hand-over: LM state saved between two evaluations, restored by a workgroup). The chains are the same chains in every form:
    iteration counts, models, poses and optical flow must equal the oracle's bit for bit in each, Huber included."""
    rows, cols, L, n = 120, 160, 4, 24
    intr = O.scaled_intrinsics(rows, cols)
    for huber in (0.0, 10.0):
        kg, kd, cg, cd, gt = synth(n, rows, cols, intr, 0x5EED9100 + mode, blocky=(mode == 2))
        ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode, huber_delta=huber), kg, kd, cg)
        forms = {"one wavefront per pair": {"VORS_REF_COOP": "0"},
                 "workgroup of 2": {"VORS_REF_COOP": "2"}, "workgroup of 3": {"VORS_REF_COOP": "3"}, "workgroup of 4": {"VORS_REF_COOP": "4"},
                 "workgroup of 5": {"VORS_REF_COOP": "5"}, "workgroup of 8": {"VORS_REF_COOP": "8"},
                 "hand-over after 25 % to workgroups of 4": {"VORS_REF_COOP": "0", "VORS_REF_HANDOFF_MIN_PAIRS": "1", "VORS_REF_HANDOFF": "25"},
                 "hand-over after 1 pair to workgroups of 2": {"VORS_REF_COOP": "0", "VORS_REF_HANDOFF_MIN_PAIRS": "1", "VORS_REF_HANDOFF": "5",
                                                               "VORS_REF_HANDOFF_WAVES": "2"}}
        for name, env in forms.items():
            for k in ("VORS_REF_COOP", "VORS_REF_HANDOFF_MIN_PAIRS", "VORS_REF_HANDOFF", "VORS_REF_HANDOFF_WAVES"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            b, poses, status, stats, _ = run_batch(vcfg(L, intr, mode, huber=huber), kg, kd, cg)
            assert_pairs_identical(ref, poses, status, stats, L, f"{MODES[mode]}, huber {huber}, {name}")
