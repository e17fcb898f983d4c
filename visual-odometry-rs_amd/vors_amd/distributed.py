"""Multi-GPU plumbing: frame pairs are independent, so they shard by contiguous blocks, one process per GPU, with no
data-path collective; the only exchange is ONE all-gather of the poses (7 f32 per pair) per step — RCCL over xGMI when
the process group backend is "nccl" (that IS RCCL on ROCm), gloo on CPU in the tests."""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous block of pairs owned by `rank`: pair i -> rank floor(i / ceil(n/world))."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def gather_poses(local_poses, out=None, group=None):
    """All ranks contribute their [P, 7] poses; every rank receives [world*P, 7] in rank order (P equal on all ranks)."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local_poses.shape[0],) + tuple(local_poses.shape[1:]), dtype=local_poses.dtype,
                          device=local_poses.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, local_poses.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out, local_poses.contiguous(), group=group)
    return out
