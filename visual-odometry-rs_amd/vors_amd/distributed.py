"""Multi-GPU plumbing: frame pairs are independent (reference: Tracker state is per instance, inverse_compositional.rs:31-34), so
they shard by contiguous blocks, one process per GPU, with no data-path collective; the only exchange is ONE all-gather of
8 f32 per pair (pose 7 + status) per step — RCCL over xGMI when the process group backend is "nccl" (that IS
RCCL on ROCm), gloo on CPU in the tests. The C ABI offers the same thing without torch: vors_multi_* (include/vors_hip.h)."""
import torch
import torch.distributed as dist


def shard_size(n_total, world):
    """Pairs per rank: ceil(n / world). Every rank's block is PADDED to this size in the collective (equal contributions)."""
    return (n_total + world - 1) // world


def shard_range(n_total, rank, world):
    """Contiguous block of pairs owned by `rank`: pair i -> rank floor(i / ceil(n/world)). The last ranks may
    own fewer pairs (or none) when world does not divide n_total; gather_results pads and trims accordingly."""
    per = shard_size(n_total, world)
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def pack_results(poses7, status, per):
    """[P,7] f32 + [P] i32 -> [per, 8] f32 (status exactly representable), rows >= P padded with status -1."""
    out = torch.zeros((per, 8), dtype=torch.float32, device=poses7.device)
    out[:, 7] = -1.0
    p = poses7.shape[0]
    if p:
        out[:p, :7] = poses7
        out[:p, 7] = status.to(torch.float32)
    return out


def gather_packed(packed, out=None, group=None):
    """All ranks contribute an equal-sized [per, 8] block; every rank receives [world*per, 8] in rank order."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device)
    if dist.get_backend(group) == "gloo":  # gloo has no device all_gather: stage through the host (tests only)
        src = packed.contiguous().cpu()
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src, group=group)
        out.copy_(torch.cat(parts, dim=0))
    else:
        dist.all_gather_into_tensor(out, packed.contiguous(), group=group)
    return out


def gather_results(poses7, status, n_total, group=None, out=None):
    """The single collective of a step: every rank passes the poses / statuses of ITS shard_range block and receives the
    [n_total, 7] poses and [n_total] statuses of the whole batch in pair order (padding trimmed)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = shard_size(n_total, world)
    lo, hi = shard_range(n_total, rank, world)
    if poses7.shape[0] != hi - lo or status.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} owns pairs [{lo}, {hi}) but passed {poses7.shape[0]} poses / {status.shape[0]} statuses")
    g = gather_packed(pack_results(poses7, status, per), out=out, group=group)
    blocks = g.view(world, per, 8)
    keep = [blocks[r, :shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]] for r in range(world)]
    flat = torch.cat(keep, dim=0)
    return flat[:, :7].contiguous(), flat[:, 7].to(torch.int32)


def gather_poses(local_poses, out=None, group=None):
    """Equal-P fast path used by bench.py (weak scaling: P pairs on every rank): [P, C] -> [world*P, C]."""
    return gather_packed(local_poses, out=out, group=group)
