"""World-size-2 tests of the N>1 path: sharding by contiguous blocks + the single all-gather of (pose 7 + status) per pair.

CPU (gloo): the host logic with stand-in results, including uneven and empty shards.
GPU (gloo control plane, both ranks on cuda:0): the REAL engine — each rank runs its own vors_batch on its shard_range of the
batch and the gathered result must be bit-identical to a single-process batch over all pairs. This is bench.py's N>1 code path
minus RCCL (the driver's 8-GPU node runs it with backend "nccl").
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vors_amd.distributed import gather_poses, gather_results, shard_range, shard_size


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(target, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    # stand-in for the per-rank HIP batch: pose of global pair i encodes i, status = i % 2 (the kernels are covered by the gpu tests)
    local = torch.stack([torch.full((7,), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros((0, 7))
    status = torch.tensor([i % 2 for i in range(lo, hi)], dtype=torch.int32)
    poses, st = gather_results(local, status, n_total)
    equal = gather_poses(torch.full((3, 8), float(rank)))   # bench.py's equal-P fast path
    # timing protocol of bench.py: barrier, then MAX over ranks of the elapsed time
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, lo, hi, poses.numpy().copy(), st.numpy().copy(), equal.numpy().copy(), float(t.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 10), (2, 7), (3, 4), (2, 1)])
def test_shard_and_gather(world, n_total):
    res = _spawn(_worker, world, n_total)
    per = shard_size(n_total, world)
    assert [(r[1], r[2]) for r in res] == [(min(r * per, n_total), min((r + 1) * per, n_total)) for r in range(world)]
    for r in res:
        assert r[3].shape == (n_total, 7) and (r[3][:, 0] == np.arange(n_total)).all()   # all poses, in global pair order
        assert (r[4] == np.arange(n_total) % 2).all()                                     # statuses travel with them
        assert (r[5][:, 0] == np.repeat(np.arange(world), 3)).all()
        assert r[6] == float(world)                                                       # MAX over ranks


def test_gather_results_rejects_wrong_shard_size():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with pytest.raises(ValueError):
            gather_results(torch.zeros((3, 7)), torch.zeros(3, dtype=torch.int32), n_total=4)
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything_once():
    for n in (1, 7, 256, 4096, 4097):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            seen = np.zeros(n, int)
            for lo, hi in blocks:
                assert 0 <= lo <= hi <= n
                seen[lo:hi] += 1
            assert (seen == 1).all()
    assert [shard_range(4096, r, 8) for r in (0, 7)] == [(0, 512), (3584, 4096)]   # BASELINE config 4: 512 per GPU


# ---------------------------------------------------------------------------------------------------- real engine, 2 ranks, 1 GPU
def _gpu_worker(rank, world, port, n_total, mode, rows, cols, L, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vors_amd as V
    from oracle import oracle as O   # intrinsics helper + nothing else (test process)
    torch.cuda.set_device(0)
    intr = O.scaled_intrinsics(rows, cols)
    lo, hi = shard_range(n_total, rank, world)
    n = hi - lo
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode)
    # every rank renders ITS pairs (seed = global pair index), like bench.py
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED2200 + lo, n, rows, cols, intr)
    b = V.Batch(cfg, n, rows, cols)
    poses = torch.zeros((n, 7), dtype=torch.float32, device="cuda")
    status = torch.zeros(n, dtype=torch.int32, device="cuda")
    b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    all_poses, all_status = gather_results(poses, status, n_total)
    q.put((rank, all_poses.cpu().numpy(), all_status.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,n_total", [(0, 24), (1, 7)], ids=["coarse_to_fine", "dense_uneven"])
def test_two_ranks_real_engine_equals_single_process(mode, n_total):
    import vors_amd as V
    from oracle import oracle as O
    rows, cols, L = 240, 320, 5
    res = _spawn(_gpu_worker, 2, n_total, mode, rows, cols, L)
    intr = O.scaled_intrinsics(rows, cols)
    cfg = V.Config(nb_levels=L, intrinsics=V.Intrinsics(intr[:2], intr[2:4], intr[4]), candidates_mode=mode)
    kg, kd, cg, _, _ = V.synth_render_pairs(0x5EED2200, n_total, rows, cols, intr)
    b = V.Batch(cfg, n_total, rows, cols)
    poses = torch.zeros((n_total, 7), dtype=torch.float32, device="cuda")
    status = torch.zeros(n_total, dtype=torch.int32, device="cuda")
    b.track_pairs(kg, kd, cg, poses, status)
    torch.cuda.synchronize()
    want_p, want_s = poses.cpu().numpy(), status.cpu().numpy()
    for rank, got_p, got_s in res:
        assert (got_p.view(np.uint32) == want_p.view(np.uint32)).all(), f"rank {rank}: sharded poses differ from the single batch"
        assert (got_s == want_s).all()
    # and against the oracle (the checker), on a sample
    k = min(n_total, 6)
    ref = O.track_pairs(O.make_config(L, intr, candidates_mode=mode), kg[:k].cpu().numpy(), kd[:k].cpu().numpy().view(np.uint16),
                        cg[:k].cpu().numpy())
    assert np.abs(want_p[:k] - ref["poses"]).max() < 1e-4 and (want_s[:k] == ref["status"]).all()


@pytest.mark.gpu
def test_bench_two_ranks_code_path_on_one_gpu():
    """bench.py's N > 1 path (rank-dependent seeds, packing pose + status, the gather, barrier + MAX-over-ranks timing, one JSON line from
    rank 0) launched exactly as the driver launches it, with the gloo TEST backend so that two ranks can share this box's GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--pairs", "64", "--total-pairs", "96", "--candidates", "c2f", "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["pairs_per_gpu"] == 64 and d["failed_pairs"] == 0
    assert abs(d["value"] - 2 * 64 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3   # whole-job aggregate over both ranks
    # round 6: with N > 1 the one line carries BOTH readings — `weak` (what `value` is by default) and `strong` = BASELINE configs[3] as
    # written (the fixed batch of --total-pairs sharded over the ranks) — each with its own gather self-check
    assert "weak" in d["config"]["value_is"]
    wk, sg = d["weak"], d["strong"]
    assert wk["scaling"] == "weak" and wk["pairs_per_gpu"] == 64 and wk["total_pairs"] == 128 and wk["value"] == d["value"]
    assert sg["scaling"] == "strong" and sg["pairs_per_gpu"] == 48 and sg["total_pairs"] == 96 and "configs[3]" in sg["workload"]
    assert abs(sg["value"] - 96 * 3 / (sg["ms_per_step"] * 3e-3)) / sg["value"] < 1e-3    # the fixed batch per step, whatever N
    for blk in (wk, sg):
        assert blk["self_check"]["gathered_blocks_equal_owners"] and blk["self_check"]["ranks"] == 2 and blk["failed_pairs_rank0"] == 0
        # ... and the same share as a continuous feed through a ring of 3 handles per rank (vors_pipeline_*), gather included
        assert blk["pipelined"]["ring"] == 3 and blk["pipelined"]["value"] > 0
    assert d["parity_pinned"] in (True, False) and d["parity_pinned_detail"]


@pytest.mark.gpu
def test_bench_two_ranks_strong_scaling_of_a_fixed_batch():
    """BASELINE configs[3] as written is a FIXED batch sharded over the GPUs (4096 pairs over 8): `--scaling strong --total-pairs T` gives
    every rank ceil(T / N) pairs of the SAME global batch (seeds = global pair indices) and reports T pairs per step whatever N is."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "3", "--warmup", "1", "--candidates", "c2f", "--scaling", "strong", "--total-pairs", "96", "--pairs", "32", "--no-secondary", "--no-pmc",
              "--no-sequences", "--cpu-pairs", "0", "--parity-pairs", "0"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo"] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d2 = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d1 = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
    for d, n, per in ((d1, 1, 96), (d2, 2, 48)):
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["config"]["pairs_per_gpu"] == per and d["config"]["total_pairs"] == 96
        assert abs(d["value"] - 96 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3   # the fixed batch per step, whatever N
    assert d2["self_check"]["gathered_blocks_equal_owners"]
    # `value` is the strong reading here and says so; the weak one (--pairs per GPU, default 4096 -> capped by the test's small --pairs) is beside it
    assert "configs[3]" in d2["config"]["workload"] and "strong" in d2["config"]["value_is"]
    assert d2["strong"]["value"] == d2["value"] and d2["weak"]["scaling"] == "weak" and d2["weak"]["self_check"]["gathered_blocks_equal_owners"]
    assert "weak" not in d1 and "strong" not in d1   # N = 1 is unchanged


@pytest.mark.gpu
def test_bench_single_rank_json_contract():
    """The ONE JSON line of `python bench.py` (N = 1) carries every key the driver's contract names, with the contract's vocabulary
    (roofline.bound in {hbm, mfma}; cpu_baseline.kind in {reference, port}), and its numbers are self-consistent."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--pairs", "64", "--steps", "3", "--warmup", "1", "--cpu-pairs", "4",
                        "--parity-pairs", "16", "--no-sequences", "--no-pmc"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0
    assert abs(d["value"] - 64 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3
    assert "errors" not in d, d.get("errors")   # (round 6: an extra leg that fails is recorded there instead of taking the line down)
    assert d["pipelined"]["ring"] == 3 and d["pipelined"]["value"] > 0 and d["parity_pinned"] in (True, False)
    assert d["config5"]["fused"]["value"] > 0 and d["config5"]["reference"]["parity_sample"]["n_poses_bit_identical"] == d["config5"]["reference"]["parity_sample"]["sample_pairs"]
    assert d["parity"]["status_equal"] and d["parity"]["n_beyond_tol"] <= d["parity"]["n_beyond_tol_oracle_f32_vs_f64_accumulation"] + 1
