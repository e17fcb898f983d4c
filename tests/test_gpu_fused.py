"""The FUSED arithmetic mode (vors_config.arithmetic = VORS_ARITH_FUSED, include/vors_hip.h) against the oracle and against the EXACT
mode. GPU only.

EXACT evaluates every per-point expression in the reference's order (bit-identical residuals / Jacobians: tests/test_gpu_parity.py).
FUSED computes the same quantities from algebraically equivalent shorter forms (warp through H = K R K^-1 and _z K t with a hardware
reciprocal, lerp-form bilinear interpolation, factored Jacobian, FMA). The bar is the north star's: poses within 1e-4 rad / 1e-4 m of
the reference arithmetic (the oracle), statuses / candidate counts / masks identical (integer stages do not depend on the mode).
FUSED-vs-EXACT differences are printed; they are of the size of the summation-order effect (tests/test_oracle_sensitivity.py).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vors_amd as V
from oracle import oracle as O

POSE_TOL = 1e-4
BLOCKY = 1 << 63


def vcfg(L, intr, mode, arith, huber=0.0, thresh=7):
    return V.Config(nb_levels=L, candidates_diff_threshold=thresh, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]),
                    candidates_mode=mode, huber_delta=huber, arithmetic=arith)


def run(cfg, kg, kd, cg):
    import torch
    n, rows, cols = kg.shape
    b = V.Batch(cfg, n, rows, cols)
    t = [torch.from_numpy(np.ascontiguousarray(kg)).cuda(), torch.from_numpy(np.ascontiguousarray(kd).view(np.int16)).cuda(),
         torch.from_numpy(np.ascontiguousarray(cg)).cuda()]
    poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
    status = torch.zeros(n, dtype=torch.int32, device="cuda")
    stats = V.stats_tensor(n)
    b.track_pairs(*t, poses, status, stats)
    torch.cuda.synchronize()
    return poses.cpu().numpy(), status.cpu().numpy(), V.decode_stats(stats)


@pytest.mark.parametrize("rows,cols,L,n,mode,huber", [
    (120, 160, 4, 16, 1, 0.0), (240, 320, 5, 8, 1, 0.0), (480, 640, 6, 6, 1, 0.0), (101, 135, 3, 4, 1, 0.0), (66, 130, 2, 3, 1, 0.0),
    (384, 512, 8, 2, 1, 0.0), (97, 131, 3, 6, 1, 8.0), (240, 320, 5, 6, 1, 10.0),
    (120, 160, 4, 16, 0, 0.0), (480, 640, 6, 6, 0, 0.0), (97, 131, 3, 6, 0, 8.0), (64, 64, 1, 2, 0, 0.0),
    (240, 320, 5, 3, 2, 0.0), (120, 160, 4, 3, 2, 0.0)])
def test_fused_vs_oracle_and_exact(rows, cols, L, n, mode, huber):
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, _, _ = O.synth_batch(n, rows, cols, seed0=(BLOCKY if mode == 2 else 0) | (0x5EEDF500 + rows), intr=intr)
    ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode, huber_delta=huber), kg, kd, cg, n_threads=8)
    pe, se, ste = run(vcfg(L, intr, mode, V.ARITH_EXACT, huber), kg, kd, cg)
    pf, sf, stf = run(vcfg(L, intr, mode, V.ARITH_FUSED, huber), kg, kd, cg)
    assert (sf == ref["status"]).all() and (se == ref["status"]).all()
    assert (stf["n_points"][:, :L] == ref["n_points"]).all()
    ok = ref["status"] == 0
    ef = np.abs(pf - ref["poses"]).max(axis=1)
    ee = np.abs(pe - ref["poses"]).max(axis=1)
    fe = np.abs(pf - pe).max(axis=1)
    print(f"[{cols}x{rows} L{L} mode{mode} huber{huber}] vs oracle: fused {ef[ok].max():.2e} exact {ee[ok].max():.2e}; fused vs exact {fe[ok].max():.2e}")
    assert (ef[ok] < POSE_TOL).all(), f"fused pose error vs oracle {ef}"
    assert (pf[~ok] == ref["poses"][~ok]).all()
    assert np.abs(stf["optical_flow"][ok] - ref["flow"][ok]).max() < 1e-3


def test_fused_negative_focal_and_skew():
    rows, cols, L = 96, 128, 3
    intr = (63.4, 47.3, 96.2, -96.0, 0.3)   # negative fv like INTRINSICS_ICL_NUIM (tum_rgbd.rs:25), non-zero skew
    kg, kd, cg, _, _ = O.synth_batch(4, rows, cols, seed0=0x5EED9000, intr=O.scaled_intrinsics(rows, cols))
    for mode in (0, 1):
        ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg, kd, cg)
        pf, sf, _ = run(vcfg(L, intr, mode, V.ARITH_FUSED), kg, kd, cg)
        assert (sf == ref["status"]).all()
        ok = sf == 0
        assert ok.any() and np.abs(pf[ok] - ref["poses"][ok]).max() < POSE_TOL


def test_fused_degenerate_inputs():
    """No depth at all, flat images, depth 1 / 65535: the same statuses as the oracle, untouched poses, no NaN leaking into a pose."""
    rows, cols, L = 64, 96, 3
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, _, _ = O.synth_batch(4, rows, cols, seed0=0x5EED8000, intr=intr)
    kd[0] = 0
    kg[1] = 128
    cg[1] = 128
    kd[2] = 65535
    kd[3] = 1
    for mode in (0, 1):
        ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg, kd, cg)
        pf, sf, stf = run(vcfg(L, intr, mode, V.ARITH_FUSED), kg, kd, cg)
        assert (sf == ref["status"]).all(), (mode, sf, ref["status"])
        assert np.isfinite(pf).all()
        bad = sf != 0
        assert (pf[bad] == ref["poses"][bad]).all()
        assert (stf["n_points"][:, :L] == ref["n_points"]).all()


def test_fused_full_size_statistics_and_determinism():
    import torch
    rows, cols, L = 480, 640, 6
    intr = O.scaled_intrinsics(rows, cols)
    for mode, n in ((1, 48), (0, 192)):
        kg, kd, cg, _, gt = V.synth_render_pairs(0x5EEDC000, n, rows, cols, intr)
        out = []
        for arith in (V.ARITH_FUSED, V.ARITH_FUSED, V.ARITH_EXACT):
            b = V.Batch(vcfg(L, intr, mode, arith), n, rows, cols)
            poses = torch.zeros((n, 7), device="cuda")
            status = torch.zeros(n, dtype=torch.int32, device="cuda")
            stats = V.stats_tensor(n)
            b.track_pairs(kg, kd, cg, poses, status, stats)
            torch.cuda.synchronize()
            out.append((poses.cpu().numpy(), status.cpu().numpy(), V.decode_stats(stats)))
        assert (out[0][0].view(np.uint32) == out[1][0].view(np.uint32)).all(), "fused mode must be deterministic"
        ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg.cpu().numpy(), kd.cpu().numpy().view(np.uint16),
                            cg.cpu().numpy(), n_threads=os.cpu_count() or 1)
        err = np.abs(out[0][0] - ref["poses"]).max(axis=1)
        fe = np.abs(out[0][0] - out[2][0]).max(axis=1)
        flips = (out[0][2]["nb_iter"][:, :L] != ref["nb_iter"]).any(axis=1).mean()
        print(f"mode {mode}: fused vs oracle max {err.max():.2e} p99 {np.quantile(err, 0.99):.2e}; fused vs exact max {fe.max():.2e}; "
              f"branch-flip rate vs oracle {flips:.0%}")
        assert (out[0][1] == ref["status"]).all()
        assert err.max() < POSE_TOL and np.quantile(err, 0.99) < 3e-5
        gt_pose = np.stack([O.iso_inverse(m) for m in gt.cpu().numpy()])
        assert np.median(np.abs(out[0][0] - gt_pose).max(axis=1)) < 3e-3


@pytest.mark.parametrize("env", [{"VORS_LM_SPLIT": "0"}, {"VORS_LM_SPLIT_ROUNDS": "1"}, {"VORS_LM_SPLIT_ROUNDS": "3"},
                                 {"VORS_LM_SPLIT_LEVELS": "1"}, {"VORS_LM_SPLIT_LEVELS": "3", "VORS_LM_CHUNKS": "7"}],
                         ids=["monolithic", "rounds1", "rounds3", "one_split_level", "three_split_levels_odd_chunks"])
def test_fused_dense_scheduling_variants(env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rows, cols, L, n = 240, 320, 5, 12
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, _, _ = O.synth_batch(n, rows, cols, seed0=0x5EED7700, intr=intr, motion_scale=2.0)
    kd[3] = 0
    ref = O.track_pairs(O.make_config(L, intr, candidates_mode=1), kg, kd, cg, n_threads=8)
    pf, sf, stf = run(vcfg(L, intr, 1, V.ARITH_FUSED), kg, kd, cg)
    assert (sf == ref["status"]).all()
    ok = sf == 0
    assert np.abs(pf[ok] - ref["poses"][ok]).max() < POSE_TOL
    assert (pf[~ok] == ref["poses"][~ok]).all()
    assert ((stf["nb_iter"][ok][:, :L] >= 1) & (stf["nb_iter"][ok][:, :L] <= 21)).all()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_fused_tracker_sequence(mode):
    rows, cols, L = 120, 160, 4
    intr = O.scaled_intrinsics(rows, cols)
    step = np.array([0.012, -0.006, 0.004, 0.002, -0.003, 0.001])
    seed = (BLOCKY | 77) if mode == 2 else 77
    frames = [O.synth_frame(seed, step * k, rows, cols, intr, frame_salt=k) for k in range(12)]
    ot = O.Tracker(O.make_config(L, intr, candidates_mode=mode), 0.0, frames[0][1], 0.0, frames[0][0])
    vt = vcfg(L, intr, mode, V.ARITH_FUSED).init(0.0, frames[0][1], 0.0, frames[0][0])
    switches = 0
    for k in range(1, len(frames)):
        g, d = frames[k]
        assert ot.track(0.1 * k, d, 0.1 * k, g) == vt.track(0.1 * k, d, 0.1 * k, g)
        assert np.abs(ot.current_frame()[1] - vt.current_frame()[1]).max() < POSE_TOL, f"frame {k}"
        assert ot.last()["changed_keyframe"] == bool(vt.last_stats()["change_keyframe"])
        switches += int(ot.last()["changed_keyframe"])
    assert switches >= 1


def test_fused_full_batch_parity_gate():
    """the out-of-tolerance TAIL of the headline arithmetic, at a sample size that can see it. Over 1024 full-size
    coarse-to-fine pairs (the bench's own seeds) the number of pairs beyond 1e-4 in FUSED arithmetic must not exceed the number the
    oracle puts beyond 1e-4 against its own f64-accumulation build (identical per-point arithmetic, only the order of the 29 sums
    differs — the floor for anything that does not sum in the reference's order) by more than one; same for 256 dense pairs.
    Branch points: lm_optimizer.rs:144,179."""
    import torch
    rows, cols, L = 480, 640, 6
    intr = O.scaled_intrinsics(rows, cols)
    for mode, n in ((0, 1024), (1, 256)):
        kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED0000, n, rows, cols, intr)
        poses = torch.zeros((n, 7), device="cuda")
        status = torch.zeros(n, dtype=torch.int32, device="cuda")
        b = V.Batch(vcfg(L, intr, mode, V.ARITH_FUSED), n, rows, cols)
        b.track_pairs(kg, kd, cg, poses, status)
        torch.cuda.synchronize()
        kgn, kdn, cgn = kg.cpu().numpy(), kd.cpu().numpy().view(np.uint16), cg.cpu().numpy()
        ocfg = O.make_config(L, intr, candidates_mode=mode)
        nt = min(os.cpu_count() or 1, n)
        ref = O.track_pairs(ocfg, kgn, kdn, cgn, n_threads=nt)
        ref64 = O.track_pairs(ocfg, kgn, kdn, cgn, n_threads=nt, variant="acc64")
        assert (status.cpu().numpy() == ref["status"]).all()
        err = np.abs(poses.cpu().numpy() - ref["poses"]).max(axis=1)
        err64 = np.abs(ref64["poses"] - ref["poses"]).max(axis=1)
        n_gpu, n_64 = int((err > POSE_TOL).sum()), int((err64 > POSE_TOL).sum())
        print(f"mode {mode}: {n} pairs, beyond 1e-4: FUSED {n_gpu}, oracle f32 vs f64 accumulation {n_64}; p99 {np.quantile(err, 0.99):.2e} max {err.max():.2e}")
        assert n_gpu <= n_64 + 1, f"mode {mode}: {n_gpu} FUSED pairs beyond 1e-4 vs {n_64} for the oracle's own summation-order probe"
        assert np.quantile(err, 0.99) < 2e-5
        del b


def test_exact_and_fused_gated_against_the_reference_arithmetic_full_batches_all_three_modes():
    """The tail of EXACT and FUSED measured against VORS_ARITH_REFERENCE — the device path that equals the oracle bit for bit
    (tests/test_gpu_reference.py) — at the bench's own sizes: 4096 coarse-to-fine pairs, 4096 DSO pairs (which the round-3 gate left
    out), 1024 dense pairs, all on the device. REFERENCE is first pinned to the oracle on the WHOLE of each batch (bits). Gates, relative to
    what was measured: EXACT and FUSED keep the statuses; each puts no more pairs beyond 1e-4 than the oracle's own summation-order floor
    (oracle f32 vs its f64-accumulation build on the same batch) + 2 + two standard deviations, per mode and over the three batches
    together — a regression that doubles the tail is red; p99 < 2e-5; nothing lands further than 5e-3 (a forked path ends at most there)."""
    import torch
    rows, cols, L = 480, 640, 6
    intr = O.scaled_intrinsics(rows, cols)
    total = {"floor": 0, "EXACT": 0, "FUSED": 0}
    for mode, n in ((0, 4096), (2, 4096), (1, 1024)):
        seed = (BLOCKY if mode == 2 else 0) | 0x5EED0000
        kg, kd, cg, _, _ = V.synth_render_pairs(seed, n, rows, cols, intr)
        res = {}
        for arith in (V.ARITH_REFERENCE, V.ARITH_EXACT, V.ARITH_FUSED):
            poses = torch.zeros((n, 7), device="cuda")
            status = torch.zeros(n, dtype=torch.int32, device="cuda")
            stats = V.stats_tensor(n)
            b = V.Batch(vcfg(L, intr, mode, arith), n, rows, cols)
            b.track_pairs(kg, kd, cg, poses, status, stats)
            torch.cuda.synchronize()
            res[arith] = (poses.cpu().numpy(), status.cpu().numpy(), V.decode_stats(stats))
            del b
        # the oracle on the WHOLE batch, in f32 (what REFERENCE must equal bit for bit) and with f64 accumulation of the 29 sums: the pairs
        # on which those two differ by more than 1e-4 are the floor no arithmetic that sums in another order can go below
        kgn, kdn, cgn = kg.cpu().numpy(), kd.cpu().numpy().view(np.uint16), cg.cpu().numpy()
        nt = min(os.cpu_count() or 1, n)
        ocfg = O.make_config(L, intr, candidates_mode=mode)
        ref = O.track_pairs(ocfg, kgn, kdn, cgn, n_threads=nt)
        ref64 = O.track_pairs(ocfg, kgn, kdn, cgn, n_threads=nt, variant="acc64")
        floor = int((np.abs(ref64["poses"] - ref["poses"]).max(axis=1) > POSE_TOL).sum())
        pr, sr, str_ = res[V.ARITH_REFERENCE]
        assert (pr.view(np.uint32) == ref["poses"].view(np.uint32)).all() and (str_["nb_iter"][:, :L] == ref["nb_iter"]).all()
        beyond = {}
        for arith, name in ((V.ARITH_EXACT, "EXACT"), (V.ARITH_FUSED, "FUSED")):
            p, st, stt = res[arith]
            assert (st == sr).all()
            err = np.abs(p - pr).max(axis=1)
            beyond[name] = int((err > POSE_TOL).sum())
            flips = (stt["nb_iter"][:, :L] != str_["nb_iter"][:, :L]).any(axis=1).mean()
            print(f"mode {mode}: {name} vs REFERENCE over {n} pairs: beyond 1e-4 {beyond[name]}, p99 {np.quantile(err, 0.99):.2e}, max {err.max():.2e}, "
                  f"LM paths that differ {flips:.0%}")
            assert np.quantile(err, 0.99) < 2e-5 and err.max() < 5e-3
            # measured (round 5, three draws of each batch): the counts scatter around the floor like independent Poisson draws of the same
            # mean (coarse-to-fine 2 / 3 vs 2, DSO 14 / 10 vs 12, dense 0 / 0 vs 0 on this seed) — two standard deviations of slack + 2
            gate = floor + 2 + 2 * np.sqrt(floor)
            print(f"mode {mode}: {name} {beyond[name]} beyond 1e-4, oracle f64-accumulation floor {floor}, gate {gate:.1f}")
            assert beyond[name] <= gate, f"mode {mode}: {beyond[name]} {name} pairs beyond 1e-4, floor {floor}"
            total[name] += beyond[name]
        total["floor"] += floor
    # ... and over the three batches together (9216 pairs), where a doubled tail cannot hide in the counting noise
    gate = total["floor"] + 2 + 2 * np.sqrt(total["floor"])
    print(f"all modes: EXACT {total['EXACT']}, FUSED {total['FUSED']}, floor {total['floor']}, gate {gate:.1f}")
    assert total["EXACT"] <= gate and total["FUSED"] <= gate, total
