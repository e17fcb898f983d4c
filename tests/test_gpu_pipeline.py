"""vors_pipeline_* (throughput mode: a ring of batch handles on internal streams): every step gives what a plain vors_batch_track_pairs
gives, bit for bit, whatever the depth and however submits, waits and drains interleave; argument checks."""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-odometry-rs_amd"))
from oracle import oracle as O
import vors_amd as V

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arith", [V.ARITH_FUSED, V.ARITH_REFERENCE], ids=["fused", "reference"])
@pytest.mark.parametrize("mode", [0, 1, 2], ids=["coarse_to_fine", "dense", "dso"])
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_pipeline_steps_equal_plain_batch_steps_bit_for_bit(mode, depth, arith):
    import torch
    rows, cols, L, n, steps = 120, 160, 4, 24, 7
    intr = O.scaled_intrinsics(rows, cols)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)
    data = []
    for k in range(steps):
        m = n - 3 * (k % 3)   # ragged feed: the batches of a feed need not be full
        kg, kd, cg, _, _ = V.synth_render_pairs((0x5EEDC000 + 97 * k) | ((1 << 63) if mode == 2 else 0), m, rows, cols, intr)
        data.append((kg, kd, cg))
    ref = V.Batch(cfg, n, rows, cols)
    want = []
    for kg, kd, cg in data:
        m = kg.shape[0]
        poses = torch.zeros((m, 7), device="cuda"); status = torch.full((m,), -9, dtype=torch.int32, device="cuda"); stats = V.stats_tensor(m)
        ref.track_pairs(kg, kd, cg, poses, status, stats)
        torch.cuda.synchronize()
        want.append((poses.cpu().numpy(), status.cpu().numpy(), V.decode_stats(stats)["nb_iter"]))
    pipe = V.Pipeline(cfg, n, rows, cols, depth=depth)
    outs, tickets = [], []
    for k, (kg, kd, cg) in enumerate(data):
        m = kg.shape[0]
        poses = torch.zeros((m, 7), device="cuda"); status = torch.full((m,), -9, dtype=torch.int32, device="cuda"); stats = V.stats_tensor(m)
        tickets.append(pipe.submit(kg, kd, cg, poses, status, stats))
        outs.append((poses, status, stats))
        if k == 2:
            pipe.wait(tickets[0])            # stream-ordered wait in the middle of the feed
            assert (outs[0][0].cpu().numpy() == want[0][0]).all()
        if k == 4:
            pipe.wait(tickets[3], host=True)  # host wait: the buffers can be read without touching the stream
    assert tickets == list(range(steps))
    pipe.drain()
    torch.cuda.synchronize()
    for k in range(steps):
        assert (outs[k][0].cpu().numpy() == want[k][0]).all(), f"poses of step {k}"
        assert (outs[k][1].cpu().numpy() == want[k][1]).all(), f"status of step {k}"
        assert (V.decode_stats(outs[k][2])["nb_iter"] == want[k][2]).all()
    with pytest.raises(V.VorsError):
        pipe.wait(steps)       # no such ticket
    with pytest.raises(V.VorsError):
        pipe.wait(-1)


@pytest.mark.parametrize("mode", [0, 2], ids=["coarse_to_fine", "dso"])
def test_a_ring_slot_picks_another_workgroup_size_and_the_same_bits(mode):
    """Round 6: a slot of a ring sizes the REFERENCE workgroup-per-pair kernel for the pairs that are resident TOGETHER (engine.h
    Geom::ref_inflight_x2): at 512 pairs per step 4 wavefronts per pair instead of the lone step's 5. The sums do not depend on the number of
    producers, so every pose must equal the plain handle's — and the oracle's — bit for bit."""
    import torch
    rows, cols, L, n = 240, 320, 5, 512
    intr = O.scaled_intrinsics(rows, cols)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=V.ARITH_REFERENCE)
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EEDC800 | ((1 << 63) if mode == 2 else 0), n, rows, cols, intr)
    ref = V.Batch(cfg, n, rows, cols)
    poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda"); stats = V.stats_tensor(n)
    ref.track_pairs(kg, kd, cg, poses, status, stats)
    torch.cuda.synchronize()
    pipe = V.Pipeline(cfg, n, rows, cols, depth=3)
    outs = []
    for k in range(4):
        p = torch.zeros((n, 7), device="cuda"); s = torch.zeros(n, dtype=torch.int32, device="cuda"); st = V.stats_tensor(n)
        pipe.submit(kg, kd, cg, p, s, st)
        outs.append((p, s, st))
    pipe.drain()
    torch.cuda.synchronize()
    want_p, want_s, want_it = poses.cpu().numpy().view(np.uint32), status.cpu().numpy(), V.decode_stats(stats)["nb_iter"]
    for p, s, st in outs:
        assert (p.cpu().numpy().view(np.uint32) == want_p).all() and (s.cpu().numpy() == want_s).all()
        assert (V.decode_stats(st)["nb_iter"] == want_it).all()
    k = 24  # ... and the oracle's, on a sample
    o = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg[:k].cpu().numpy(), kd[:k].cpu().numpy().view(np.uint16), cg[:k].cpu().numpy(), n_threads=8)
    assert (want_p[:k] == o["poses"].view(np.uint32)).all()


def test_pipeline_argument_checks_and_inputs_ordered_on_the_callers_stream():
    import torch
    rows, cols, L, n = 60, 80, 3, 4
    intr = O.scaled_intrinsics(rows, cols)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]))
    for bad_depth in (0, 9):
        with pytest.raises(V.VorsError):
            V.Pipeline(cfg, n, rows, cols, depth=bad_depth)
    with pytest.raises(V.VorsError):
        V.Pipeline(cfg, n, rows, cols, device=torch.cuda.device_count())
    pipe = V.Pipeline(cfg, n, rows, cols, depth=2)
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EEDC100, n, rows, cols, intr)
    poses = torch.zeros((n, 7), device="cuda"); status = torch.zeros(n, dtype=torch.int32, device="cuda")
    with pytest.raises(V.VorsError):
        pipe.submit(kg[:, :-1], kd, cg, poses, status)
    # the inputs are produced on the caller's stream right before the submit (a copy from a staging tensor): the step must see them
    ref = V.Batch(cfg, n, rows, cols)
    want = torch.zeros((n, 7), device="cuda"); wstat = torch.zeros(n, dtype=torch.int32, device="cuda")
    ref.track_pairs(kg, kd, cg, want, wstat)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        kg2 = torch.zeros_like(kg); kd2 = torch.zeros_like(kd); cg2 = torch.zeros_like(cg)
        for _ in range(20):                      # some queued work in front of the copies
            torch.zeros((1 << 20,), device="cuda").sum()
        kg2.copy_(kg); kd2.copy_(kd); cg2.copy_(cg)
        t = pipe.submit(kg2, kd2, cg2, poses, status)
        pipe.wait(t)
        got = poses.clone()
    s.synchronize()
    assert (got.cpu().numpy() == want.cpu().numpy()).all()
