"""Multi-GPU partitioning of the analysis batch.

Every unit of work (a sequence for K2-K4, a matrix for K5) is independent, so the batch is cut into contiguous
per-rank slabs with no data-path collective; the only exchange is ONE all-gather of the per-rank spectra slabs
(RCCL over xGMI when the process group's backend is "nccl"; gloo on CPU in the tests).  The reference has no
distributed code at all (single process, device "cuda": launch.py:63-66, analysis/eval_eig.py:497).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n, rank=None, world_size=None):
    """Contiguous slab [lo, hi) of n units for `rank`: ceil(n / world) per rank, the tail ranks may be short or empty."""
    if rank is None or world_size is None:
        rank, world_size = world()
    per = (n + world_size - 1) // world_size
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi


def all_gather_slabs(local, n_total, group=None):
    """Collect per-rank slabs (cut by shard_range along dim 0) into the full [n_total, ...] tensor on every rank
    with a single all_gather.  Slabs are padded to the common ceil(n/world) length so one fixed-size collective
    suffices; the padding is dropped on return."""
    rank, world_size = world()
    if world_size == 1:
        return local
    per = (n_total + world_size - 1) // world_size
    pad = per - local.shape[0]
    if pad > 0:
        filler = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, filler], dim=0)
    local = local.contiguous()
    out = torch.empty((world_size * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_complex():
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(local), group=group)
    else:
        dist.all_gather_into_tensor(out, local, group=group)
    return out[:n_total]
