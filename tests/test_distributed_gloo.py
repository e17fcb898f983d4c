"""World-size-2 gloo test of the N>1 path's host logic (sharding by contiguous blocks + the single pose all-gather)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vors_amd.distributed import gather_poses, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    # stand-in for the per-rank HIP batch: pose of global pair i encodes i (the kernels are covered by the gpu tests)
    local = torch.stack([torch.full((7,), float(i)) for i in range(lo, hi)])
    out = gather_poses(local)
    # timing protocol of bench.py: barrier, then MAX over ranks of the elapsed time
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, lo, hi, out.numpy().copy(), float(t.item())))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world, n_total = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 5), (5, 10)]
    for r in res:
        assert (r[3][:, 0] == np.arange(n_total)).all()   # every rank sees all poses, in global pair order
        assert r[4] == 2.0                                 # MAX over ranks


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
