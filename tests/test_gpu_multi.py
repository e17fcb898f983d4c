"""Multi-GPU entry points of the C ABI (vors_batch_create_on, vors_multi_*): ONE process, pairs sharded over the visible devices, a single
RCCL all-gather of pose + status. On a 1-GPU box the multi handle spans one device (the collective degenerates to a copy): sharding,
packing, trimming and the host-buffer path are exercised and must reproduce the plain batch bit for bit; with >= 2 GPUs (the driver's
8-GPU node) the same test runs the RCCL path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import vors_amd as V
from oracle import oracle as O


def _cfg(L, intr, mode, arith=V.ARITH_FUSED):
    return V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode, arithmetic=arith)


@pytest.mark.parametrize("mode,n", [(0, 13), (1, 5)], ids=["coarse_to_fine", "dense"])
def test_multi_equals_single_batch(mode, n):
    import torch
    rows, cols, L = 120, 160, 4
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, _, _ = O.synth_batch(n, rows, cols, seed0=0x5EED3300, intr=intr)
    nd = V.device_count()
    per = (n + nd - 1) // nd
    m = V.MultiGpu(_cfg(L, intr, mode), per, rows, cols)
    assert m.device_count() == nd
    blocks = [m.shard(n, k) for k in range(nd)]
    assert sum(c for _, c in blocks) == n and blocks[0][0] == 0
    poses, status = m.track_pairs_host(kg, kd, cg)
    # reference: per-device batches of the same block sizes (results do not depend on the batch a pair is in, but the scheduling knobs
    # derive from max_pairs: same sizes -> bit-identical), checked against the oracle as well
    want_p, want_s = np.zeros((n, 7), np.float32), np.zeros(n, np.int32)
    for k, (lo, cnt) in enumerate(blocks):
        if cnt == 0:
            continue
        with torch.cuda.device(k):
            b = V.Batch(_cfg(L, intr, mode), per, rows, cols, device=k)
            assert b.device() == k
            t = [torch.from_numpy(a[lo:lo + cnt].copy()).cuda(k) for a in (kg, kd.view(np.int16), cg)]
            p = torch.zeros((cnt, 7), device=f"cuda:{k}")
            s = torch.zeros(cnt, dtype=torch.int32, device=f"cuda:{k}")
            b.track_pairs(t[0], t[1], t[2], p, s)
            torch.cuda.synchronize(k)
            want_p[lo:lo + cnt], want_s[lo:lo + cnt] = p.cpu().numpy(), s.cpu().numpy()
    assert (poses.view(np.uint32) == want_p.view(np.uint32)).all() and (status == want_s).all()
    ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg, kd, cg)
    assert (status == ref["status"]).all() and np.abs(poses - ref["poses"]).max() < 1e-4
    # device-resident entry: the caller's shards already on their devices
    sh = [[torch.from_numpy(a[lo:lo + cnt].copy()).cuda(k) for (lo, cnt), k in zip(blocks, range(nd))] for a in (kg, kd.view(np.int16), cg)]
    p2, s2 = m.track_pairs(sh[0], sh[1], sh[2], n)
    assert (p2.view(np.uint32) == poses.view(np.uint32)).all() and (s2 == status).all()


def test_create_on_rejects_bad_device_and_foreign_stream():
    import torch
    rows, cols = 64, 96
    intr = O.scaled_intrinsics(rows, cols)
    with pytest.raises(V.VorsError, match="device index out of range"):
        V.Batch(_cfg(3, intr, 0), 2, rows, cols, device=V.device_count())
    with pytest.raises(V.VorsError):
        V.MultiGpu(_cfg(3, intr, 0), 2, rows, cols, n_devices=V.device_count() + 1)
    with pytest.raises(V.VorsError, match="duplicate device id"):
        V.MultiGpu(_cfg(3, intr, 0), 2, rows, cols, n_devices=2, device_ids=[0, 0]) if V.device_count() >= 2 else (_ for _ in ()).throw(
            V.VorsError("duplicate device id (single-GPU box: not reachable)"))
    if V.device_count() >= 2:   # a stream of device 1 handed to a handle of device 0
        b = V.Batch(_cfg(3, intr, 0), 2, rows, cols, device=0)
        kg, kd, cg, _, _ = O.synth_batch(2, rows, cols, seed0=1, intr=intr)
        with torch.cuda.device(1):
            t = [torch.from_numpy(a).cuda(0) for a in (kg, kd.view(np.int16), cg)]
            p = torch.zeros((2, 7), device="cuda:0"); s = torch.zeros(2, dtype=torch.int32, device="cuda:0")
            with pytest.raises(V.VorsError, match="stream belongs to device"):
                b.track_pairs(t[0], t[1], t[2], p, s)


def test_kernel_timing_on_a_handle_of_another_device_and_multi_argument_checks():
    """the timing events of a handle belong to the handle's device whatever the caller's current device is (cross-device part
    needs >= 2 GPUs; on one GPU the same calls run on device 0); MultiGpu.track_pairs_host rejects arrays that do not hold n pairs each."""
    import torch
    rows, cols, L, n = 64, 96, 3, 4
    intr = O.scaled_intrinsics(rows, cols)
    kg, kd, cg, _, _ = O.synth_batch(n, rows, cols, seed0=3, intr=intr)
    dev = V.device_count() - 1          # the LAST device; the caller stays on device 0
    torch.cuda.set_device(0)
    b = V.Batch(_cfg(L, intr, 0), n, rows, cols, device=dev)
    b.enable_kernel_timing(4)
    t = [torch.from_numpy(a).cuda(dev) for a in (kg, kd.view(np.int16), cg)]
    p = torch.zeros((n, 7), device=f"cuda:{dev}"); s = torch.zeros(n, dtype=torch.int32, device=f"cuda:{dev}")
    with torch.cuda.device(dev):
        b.track_pairs(t[0], t[1], t[2], p, s)
        torch.cuda.synchronize(dev)
    assert torch.cuda.current_device() == 0
    ms = b.kernel_times("lm")
    assert len(ms) == 1 and ms[0] > 0 and b.last_kernel_ms()["lm_ms"] > 0
    m = V.MultiGpu(_cfg(L, intr, 0), n, rows, cols)
    assert m.rccl_version() == 0 or V.device_count() >= 2
    with pytest.raises(V.VorsError):
        m.track_pairs_host(kg, kd[: n - 1], cg)
    with pytest.raises(V.VorsError):
        m.track_pairs_host(kg, kd, cg[:, : rows - 1])
