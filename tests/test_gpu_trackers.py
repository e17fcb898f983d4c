"""vors_trackers_*: N sequences advancing in lock-step with the tracker state machine on the device (per-sequence keyframe promotion
through masked launches) must equal N single vors_tracker handles bit for bit, and the oracle's Tracker within the pose tolerance.
Reference: src/bin/vors_track.rs:46-62, src/core/track/inverse_compositional.rs:170-240. GPU only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vors_amd as V
from oracle import oracle as O

POSE_TOL = 1e-4
BLOCKY = 1 << 63


def make_sequences(n_seq, n_frames, rows, cols, intr, blocky):
    """n_seq sequences of one scene family: sequence s moves along its own twist direction at its own speed, so the optical-flow
    threshold (inverse_compositional.rs:224) is crossed at different frames in different sequences. -> gray [F][n_seq,...], depth."""
    import torch
    base = np.array([0.012, -0.006, 0.004, 0.002, -0.003, 0.001])
    rng = np.random.default_rng(7)
    speed = 0.35 + 1.3 * rng.random(n_seq)           # some sequences switch keyframes every 2-3 frames, some hardly ever
    sign = rng.choice([-1.0, 1.0], size=(n_seq, 6))
    frames = []
    for k in range(n_frames):
        seeds = [(BLOCKY if blocky else 0) | (1000 + s) for s in range(n_seq)]
        salts = [k] * n_seq
        xis = [base * sign[s] * speed[s] * k for s in range(n_seq)]
        g, d = V.synth_render_frames(seeds, salts, xis, rows, cols, intr)
        frames.append((g, d))
    torch.cuda.synchronize()
    return frames


@pytest.mark.parametrize("arith", [V.ARITH_EXACT, V.ARITH_FUSED], ids=["exact", "fused"])
@pytest.mark.parametrize("mode,rows,cols,L", [(0, 120, 160, 4), (1, 120, 160, 4), (2, 120, 160, 4)], ids=["coarse_to_fine", "dense", "dso"])
def test_64_sequences_equal_64_single_trackers_bit_for_bit(mode, rows, cols, L, arith):
    n_seq, n_frames = 64, 9
    intr = O.scaled_intrinsics(rows, cols)
    frames = make_sequences(n_seq, n_frames, rows, cols, intr, blocky=(mode == 2))
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)
    host = [(g.cpu().numpy(), d.cpu().numpy().view(np.uint16)) for g, d in frames]

    many = V.Trackers(cfg, n_seq, rows, cols)
    many.init(*frames[0])
    singles = [cfg.init(0.0, host[0][1][s], 0.0, host[0][0][s]) for s in range(n_seq)]
    n_switch = np.zeros(n_seq, int)
    for k in range(1, n_frames):
        many.track(*frames[k])
        poses, status, kf_index = many.current_frames()
        st_many = many.stats()
        for s in range(n_seq):
            st1 = singles[s].track(float(k), host[k][1][s], float(k), host[k][0][s])
            p1 = singles[s].current_frame()[1]
            assert st1 == status[s], f"frame {k} sequence {s}: status"
            assert (p1.view(np.uint32) == poses[s].view(np.uint32)).all(), f"frame {k} sequence {s}: pose bits differ ({np.abs(p1 - poses[s]).max()})"
            l1 = singles[s].last_stats()
            assert int(l1["change_keyframe"]) == int(st_many[s]["change_keyframe"])
            assert (l1["nb_iter"] == st_many[s]["nb_iter"]).all()
            n_switch[s] += int(l1["change_keyframe"])
            # keyframe bookkeeping: the frame index of the keyframe = the last frame that switched
            assert kf_index[s] == (k if l1["change_keyframe"] else kf_index[s])
            assert float(kf_index[s]) == singles[s].keyframe()[0]
    # the point of the test: promotions happen at different frames in different sequences (masked launches over a partial list)
    assert len(set(n_switch.tolist())) >= 3 and n_switch.max() >= 2, f"switch counts {sorted(set(n_switch.tolist()))}"


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["coarse_to_fine", "dense", "dso"])
def test_sequences_vs_oracle_tracker(mode):
    rows, cols, L, n_seq, n_frames = 120, 160, 4, 6, 10
    intr = O.scaled_intrinsics(rows, cols)
    frames = make_sequences(n_seq, n_frames, rows, cols, intr, blocky=(mode == 2))
    host = [(g.cpu().numpy(), d.cpu().numpy().view(np.uint16)) for g, d in frames]
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_FUSED)
    many = V.Trackers(cfg, n_seq, rows, cols)
    many.init(*frames[0])
    ots = [O.Tracker(O.make_config(L, intr, candidates_mode=mode), 0.0, host[0][1][s], 0.0, host[0][0][s]) for s in range(n_seq)]
    switches = 0
    for k in range(1, n_frames):
        many.track(*frames[k])
        poses, status, kf_index = many.current_frames()
        for s in range(n_seq):
            assert ots[s].track(float(k), host[k][1][s], float(k), host[k][0][s]) == status[s]
            assert np.abs(ots[s].current_frame()[1] - poses[s]).max() < POSE_TOL, f"frame {k} sequence {s}"
            switches += int(ots[s].last()["changed_keyframe"])
            assert float(kf_index[s]) == ots[s].keyframe_pose()[0]
    assert switches >= 3


def test_trackers_argument_checks():
    import torch
    rows, cols = 60, 80
    intr = O.scaled_intrinsics(rows, cols)
    cfg = V.Config(nb_levels=3, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]))
    t = V.Trackers(cfg, 2, rows, cols)
    g = torch.zeros((2, rows, cols), dtype=torch.uint8, device="cuda")
    d = torch.zeros((2, rows, cols), dtype=torch.int16, device="cuda")
    with pytest.raises(V.VorsError):
        t.track(g, d)  # before init
    with pytest.raises(V.VorsError):
        t.current_frames()
    with pytest.raises(V.VorsError):
        t.init(g[:1], d[:1])
    t.init(g, d)
    t.track(g, d)      # no usable candidate anywhere: the pose stays the identity, like the reference
    poses, status, kf = t.current_frames()
    assert (poses == np.array([0, 0, 0, 0, 0, 0, 1], np.float32)).all()
    # the explicit-device constructor: device 0 behaves like the default, a device that does not exist is refused
    t0 = V.Trackers(cfg, 2, rows, cols, device=0)
    t0.init(g, d)
    t0.track(g, d)
    assert (t0.current_frames()[0] == poses).all()
    with pytest.raises(V.VorsError):
        V.Trackers(cfg, 2, rows, cols, device=torch.cuda.device_count())


def _dso_sequences(n_seq, n_frames, rows, cols, intr, seed0):
    """n_seq DSO-textured sequences with their own seeds, speeds and directions -> device frames [F][n_seq, rows, cols] + host copies."""
    import torch
    base = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])
    rng = np.random.default_rng(seed0)
    speed = 0.5 + 1.0 * rng.random(n_seq)
    sign = rng.choice([-1.0, 1.0], size=(n_seq, 6))
    frames = []
    for k in range(n_frames):
        frames.append(V.synth_render_frames([BLOCKY | (seed0 + s) for s in range(n_seq)], [k] * n_seq,
                                            [base * sign[s] * speed[s] * k for s in range(n_seq)], rows, cols, intr))
    torch.cuda.synchronize()
    gh = np.stack([g.cpu().numpy() for g, _ in frames])
    dh = np.stack([d.cpu().numpy().view(np.uint16) for _, d in frames])
    return frames, gh, dh


def _run_trackers(cfg, frames, n_seq, rows, cols):
    many = V.Trackers(cfg, n_seq, rows, cols)
    many.init(*frames[0])
    traj, stat, sw = [], [], []
    for k in range(1, len(frames)):
        many.track(*frames[k])
        poses, status, _ = many.current_frames()
        traj.append(poses)
        stat.append(status)
        sw.append(many.stats()["change_keyframe"].copy())
    return np.stack(traj, axis=1), np.stack(stat, axis=1), np.stack(sw, axis=1)


def test_config3_shape_640x480_dso_sequences_16_seeds_of_60_frames_vs_oracle():
    """BASELINE configs[2]'s real shape — 640x480, 6 levels, SEQUENCES with DSO candidate selection (every keyframe switch re-runs the
    selector on the frame that was current, inverse_compositional.rs:224-239) — over SIXTEEN seeds x 60 tracked frames instead of the one
    seed round 3 asserted on (a seed-selected pass). A tracked pose is a chain product, so one alignment that lands 1e-3 away
    moves every later pose of its sequence.
      REFERENCE arithmetic: every pose of every frame of every sequence equals the oracle tracker's BIT FOR BIT (and so do the keyframe
      decisions and statuses) — asserted.
      EXACT / FUSED: the number of sequences that leave the 1e-4 band anywhere along their 60 frames is REPORTED next to the oracle's own
      f32-vs-f64-accumulation count and gated against it (the LM path forks on the order of the sums); no sequence may
      drift further than 5e-3."""
    rows, cols, L, n_seq, n = 480, 640, 6, 16, 61
    intr = O.INTRINSICS_FR1
    frames, gh, dh = _dso_sequences(n_seq, n, rows, cols, intr, 31337)
    ocfg = O.make_config(L, intr, candidates_mode=V.CANDIDATES_DSO)
    ref = O.track_sequences(ocfg, gh, dh, n_threads=n_seq)
    ref64 = O.track_sequences(ocfg, gh, dh, n_threads=n_seq, variant="acc64")
    floor = int((np.abs(ref64["poses"] - ref["poses"]).max(axis=(1, 2)) > POSE_TOL).sum())
    assert ref["changed_keyframe"].sum() >= n_seq
    for arith, name in ((V.ARITH_REFERENCE, "reference"), (V.ARITH_EXACT, "exact"), (V.ARITH_FUSED, "fused")):
        cfg = V.Config(nb_levels=L, intrinsics=V.INTRINSICS_FR1, candidates_mode=V.CANDIDATES_DSO, arithmetic=arith)
        traj, stat, sw = _run_trackers(cfg, frames, n_seq, rows, cols)
        assert (stat == ref["status"]).all()
        err = np.abs(traj - ref["poses"]).max(axis=(1, 2))
        if arith == V.ARITH_REFERENCE:
            assert (traj.view(np.uint32) == ref["poses"].view(np.uint32)).all(), f"REFERENCE: max difference {err.max():.3e}"
            assert (sw == ref["changed_keyframe"]).all()
        else:
            n_out = int((err > POSE_TOL).sum())
            print(f"{name}: {n_out} of {n_seq} sequences leave the 1e-4 band within 60 frames (oracle f32 vs f64 accumulation: {floor}); max {err.max():.2e}")
            # measured on this seed (round 5): floor 1, EXACT 2, FUSED 2 — one sequence of slack, and the typical sequence stays far inside the band
            assert n_out <= floor + 1, f"{name}: {n_out} sequences beyond 1e-4, the oracle's own summation-order floor is {floor}"
            assert np.median(err) < 2e-5 and err.max() < 5e-3


def test_config3_at_full_length_600_frame_dso_sequences_reference_arithmetic_bit_identical():
    """configs[2] at the LENGTH of fr1/desk (~600 frames; the dataset itself is not available offline): 8 synthetic 640x480 DSO sequences
    of 600 tracked frames through vors_trackers_* in the REFERENCE arithmetic against the oracle tracker — 4800 chained alignments with
    ~25 keyframe switches per sequence, every pose identical bit for bit; FUSED on the same frames reported (sequences beyond 1e-4)."""
    rows, cols, L, n_seq, n = 480, 640, 6, 8, 601
    intr = O.INTRINSICS_FR1
    frames, gh, dh = _dso_sequences(n_seq, n, rows, cols, intr, 4242)
    ocfg = O.make_config(L, intr, candidates_mode=V.CANDIDATES_DSO)
    ref = O.track_sequences(ocfg, gh, dh, n_threads=n_seq)
    cfg = V.Config(nb_levels=L, intrinsics=V.INTRINSICS_FR1, candidates_mode=V.CANDIDATES_DSO, arithmetic=V.ARITH_REFERENCE)
    traj, stat, sw = _run_trackers(cfg, frames, n_seq, rows, cols)
    assert (stat == ref["status"]).all() and (sw == ref["changed_keyframe"]).all()
    assert (traj.view(np.uint32) == ref["poses"].view(np.uint32)).all(), f"max difference {np.abs(traj - ref['poses']).max():.3e}"
    assert ref["changed_keyframe"].sum() >= 5 * n_seq
    # FUSED on the same frames, gated against the oracle's own summation-order floor (f32 vs f64 accumulation over the same 600 frames):
    # a pose is a chain product, so over 600 frames most sequences cross 1e-4 in EITHER arithmetic — what is gated is that FUSED does not
    # put more sequences outside than the floor plus one, and that nothing drifts further than a forked LM path does (5e-3)
    ref64 = O.track_sequences(ocfg, gh, dh, n_threads=n_seq, variant="acc64")
    floor = int((np.abs(ref64["poses"] - ref["poses"]).max(axis=(1, 2)) > POSE_TOL).sum())
    cfg.arithmetic = V.ARITH_FUSED
    trajf, statf, _ = _run_trackers(cfg, frames, n_seq, rows, cols)
    drift = np.abs(trajf - ref["poses"]).max(axis=2)          # [n_seq, frames]
    n_out = int((drift.max(axis=1) > POSE_TOL).sum())
    print(f"FUSED over 600 frames: {n_out} of {n_seq} sequences leave the 1e-4 band (oracle f32 vs f64 accumulation: {floor}); per-frame max drift "
          f"at frames 100/300/600: {drift[:, 99].max():.2e} / {drift[:, 299].max():.2e} / {drift[:, 599].max():.2e}")
    assert (statf == ref["status"]).all()
    assert n_out <= floor + 1, f"FUSED: {n_out} of {n_seq} sequences beyond 1e-4 over 600 frames, the oracle's own floor is {floor}"
    assert drift.max() < 5e-3


@pytest.mark.parametrize("mode", [0, 1], ids=["coarse_to_fine", "dense"])
def test_sequences_failure_semantics_vs_oracle(mode):
    """The reference's failure paths inside a SEQUENCE, with the state machine on the device: a keyframe switch onto a frame without any
    valid depth leaves a keyframe with no candidates — from then on step() fails (Cholesky of a zero Hessian), the pose is KEPT
    (inverse_compositional.rs:195-199,206-208), the optical flow is 0/0 = NaN and never switches again (:213-224). Sequences 0 and 2 get
    such a frame at different times, sequence 1 never: statuses, poses and keyframe indices must follow the oracle frame by frame."""
    rows, cols, L, n_seq, n_frames = 120, 160, 4, 3, 9
    intr = O.scaled_intrinsics(rows, cols)
    frames = make_sequences(n_seq, n_frames, rows, cols, intr, blocky=False)
    host = [(g.cpu().numpy().copy(), d.cpu().numpy().view(np.uint16).copy()) for g, d in frames]
    # zero the depth of frame 3 in sequence 0 and of frames 5.. in sequence 2 (whichever of them becomes a keyframe kills the sequence)
    host[3][1][0][:] = 0
    for k in range(5, n_frames):
        host[k][1][2][:] = 0
    import torch
    dev = [(torch.from_numpy(g).cuda(), torch.from_numpy(d.view(np.int16)).cuda()) for g, d in host]
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_FUSED)
    many = V.Trackers(cfg, n_seq, rows, cols)
    many.init(*dev[0])
    ots = [O.Tracker(O.make_config(L, intr, candidates_mode=mode), 0.0, host[0][1][s], 0.0, host[0][0][s]) for s in range(n_seq)]
    failed = 0
    for k in range(1, n_frames):
        many.track(*dev[k])
        poses, status, kf_index = many.current_frames()
        for s in range(n_seq):
            ost = ots[s].track(float(k), host[k][1][s], float(k), host[k][0][s])
            assert ost == status[s], f"frame {k} sequence {s}: status {status[s]} vs oracle {ost}"
            po = ots[s].current_frame()[1]
            if ost != 0:
                failed += 1
                assert (po == poses[s]).all() or np.abs(po - poses[s]).max() < POSE_TOL, f"frame {k} sequence {s}: kept pose"
            else:
                assert np.abs(po - poses[s]).max() < POSE_TOL, f"frame {k} sequence {s}"
            assert float(kf_index[s]) == ots[s].keyframe_pose()[0], f"frame {k} sequence {s}: keyframe index"
    assert failed >= 2, "the scenario was meant to kill at least one sequence"


@pytest.mark.parametrize("arith", [V.ARITH_EXACT, V.ARITH_FUSED], ids=["exact", "fused"])
def test_dense_128_sequences_equal_single_trackers_beyond_the_chunk_cap(arith):
    """The evaluation rounds cut a level-0 evaluation into chunks whose count fixes the order of the f32 partial sums; it must be the same
    for EVERY handle below 512 sequences (it was 256 below 128 pairs and 128 from 128 on, equalised only by the S0 / 2400 cap
    up to 640x480). 640x512 dense, 128 sequences in one lock-step handle against single trackers, bit for bit."""
    rows, cols, L, n_seq, n_frames = 512, 640, 6, 128, 4
    intr = O.scaled_intrinsics(rows, cols)
    frames = make_sequences(n_seq, n_frames, rows, cols, intr, blocky=False)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=V.CANDIDATES_DENSE, arithmetic=arith)
    many = V.Trackers(cfg, n_seq, rows, cols)
    many.init(*frames[0])
    pick = [0, 37, 127]
    host = [(g[pick].cpu().numpy(), d[pick].cpu().numpy().view(np.uint16)) for g, d in frames]
    singles = [cfg.init(0.0, host[0][1][j], 0.0, host[0][0][j]) for j in range(len(pick))]
    for k in range(1, n_frames):
        many.track(*frames[k])
        poses, status, _ = many.current_frames()
        for j, s_ in enumerate(pick):
            assert singles[j].track(float(k), host[k][1][j], float(k), host[k][0][j]) == status[s_]
            p1 = singles[j].current_frame()[1]
            assert (p1.view(np.uint32) == poses[s_].view(np.uint32)).all(), f"frame {k} sequence {s_}: {np.abs(p1 - poses[s_]).max():.3e}"


def test_device_frame_renderer_agrees_with_the_cpu_renderer():
    """vors_synth_render_frames (HIP, f64) and the oracle's CPU renderer evaluate the same scene function (csrc/synth_scene.h): the frames may
    differ only where a transcendental's last bit moves a value across a rounding boundary — a handful of pixels by one grey level / one
    depth unit. (The parity tests never rely on this: they feed BOTH sides the frames one renderer made.)"""
    rows, cols = 120, 160
    intr = O.scaled_intrinsics(rows, cols)
    xis = [np.array([0.012, -0.006, 0.004, 0.002, -0.003, 0.001]) * k for k in range(4)]
    for seed in (77, BLOCKY | 77):
        g, d = V.synth_render_frames([seed] * 4, list(range(4)), xis, rows, cols, intr)
        g, d = g.cpu().numpy(), d.cpu().numpy().view(np.uint16)
        for k in range(4):
            cg, cd = O.synth_frame(seed, xis[k], rows, cols, intr, frame_salt=k)
            dg = np.abs(g[k].astype(int) - cg.astype(int))
            assert (dg > 0).mean() < 2e-3 and ((dg <= 1) | (seed >> 63 == 1)).all(), f"grey: {(dg > 0).sum()} pixels differ, max {dg.max()}"
            dd = np.abs(d[k].astype(int) - cd.astype(int))
            assert ((d[k] == 0) == (cd == 0)).all() and (dd > 0).mean() < 2e-3 and dd.max() <= 1


@pytest.mark.parametrize("mode", [V.CANDIDATES_COARSE_TO_FINE, V.CANDIDATES_DSO], ids=["coarse_to_fine", "dso"])
def test_64_sequence_tail_of_exact_and_fused_is_the_oracles_own_floor(mode):
    """bench.py's `sequences_64` statistic (64 sequences x 39 tracked frames, 640x480, 6 levels) as a gate, over TWO draws: the number of
    sequences whose worst frame lies beyond 1e-4 of the oracle tracker, for EXACT and FUSED, against the oracle's own f32-vs-f64-accumulation
    count on the same frames (tools/seq_parity.py measures the same over six draws: 9 / 384 coarse-to-fine, 43 / 384 DSO, equal to the floor
    within counting noise). Gate: beyond <= floor + 2 + 2 sqrt(floor) summed over the draws — a regression that doubles the tail is red;
    REFERENCE on the same frames: every pose bit-identical."""
    rows, cols, L, n_seq, n_frames = 480, 640, 6, 64, 40
    intr = O.scaled_intrinsics(rows, cols)
    base = np.array([0.004, -0.002, 0.0015, 0.0008, -0.001, 0.0005])
    blocky = BLOCKY if mode == V.CANDIDATES_DSO else 0
    tot = {"floor": 0, "exact": 0, "fused": 0}
    for d in range(2):
        rng = np.random.default_rng(11 + d)
        speed = 0.5 + 1.0 * rng.random(n_seq)
        sign = rng.choice([-1.0, 1.0], size=(n_seq, 6))
        frames = [V.synth_render_frames([blocky | (4242 + 1000 * d + s_) for s_ in range(n_seq)], [k] * n_seq,
                                        [base * sign[s_] * speed[s_] * k for s_ in range(n_seq)], rows, cols, intr) for k in range(n_frames)]
        gh = np.stack([g.cpu().numpy() for g, _ in frames])
        dh = np.stack([x.cpu().numpy().view(np.uint16) for _, x in frames])
        ocfg = O.make_config(L, intr, candidates_mode=mode)
        nt = min(os.cpu_count() or 1, n_seq)
        ref = O.track_sequences(ocfg, gh, dh, n_threads=nt)
        ref64 = O.track_sequences(ocfg, gh, dh, n_threads=nt, variant="acc64")
        tot["floor"] += int((np.abs(ref64["poses"] - ref["poses"]).max(axis=(1, 2)) > POSE_TOL).sum())
        for arith, name in ((V.ARITH_REFERENCE, "reference"), (V.ARITH_EXACT, "exact"), (V.ARITH_FUSED, "fused")):
            cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)
            traj, stat, _ = _run_trackers(cfg, frames, n_seq, rows, cols)
            assert (stat == ref["status"]).all()
            if arith == V.ARITH_REFERENCE:
                assert (traj.view(np.uint32) == ref["poses"].view(np.uint32)).all()
            else:
                err = np.abs(traj - ref["poses"]).max(axis=(1, 2))
                tot[name] += int((err > POSE_TOL).sum())
                assert np.median(err) < 2e-5 and err.max() < 5e-3
        del frames
    gate = tot["floor"] + 2 + 2 * np.sqrt(tot["floor"])
    print(f"mode {mode}: sequences beyond 1e-4 of 128: oracle f64-accumulation floor {tot['floor']}, EXACT {tot['exact']}, FUSED {tot['fused']} (gate {gate:.1f})")
    assert tot["exact"] <= gate and tot["fused"] <= gate, tot
