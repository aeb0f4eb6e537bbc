"""Multi-GPU partitioning of the analysis batch.

Every unit of work (a sequence for K2-K4, a matrix for K5) is independent, so the batch is cut into contiguous
per-rank slabs with no data-path collective; the only exchange is ONE all-gather of the per-rank spectra slabs
(RCCL over xGMI when the process group's backend is "nccl"; gloo on CPU in the tests).  The reference has no
distributed code at all (single process, device "cuda": launch.py:63-66, analysis/eval_eig.py:497).
"""
import os

import torch
import torch.distributed as dist


def bootstrap():
    """Join the job's process group when this process was started by a distributed launcher (``torchrun launch.py ...``:
    WORLD_SIZE / RANK / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment) and nobody has initialised one yet.
    The reference's caller (launch.py:176, :183) knows nothing about ranks, so the drop-in entry point does this itself:
    one process per GPU, device = LOCAL_RANK, backend "nccl" (= RCCL over xGMI on ROCm).  EGS_DIST_BACKEND=gloo selects
    the CPU backend (tests; several ranks sharing one GPU).  Returns (rank, world_size)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if dist.is_available() and not dist.is_initialized() and ws > 1:
        backend = os.environ.get("EGS_DIST_BACKEND", "nccl")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")), world_size=ws)
    return world()


def _env_flag(name):
    """'1' / 'true' / 'yes' / 'on' (any case) switch a flag on; unset, empty, '0', 'false', 'no', 'off' leave it off."""
    return os.environ.get(name, "").strip().lower() in ("1", "true", "yes", "on")


_gather_sync_agreed = {}


def gather_sync(group=None):
    """EGS_GATHER_SYNC, read ONCE per process group and agreed over it: ranks that disagreed would mix the blocking
    collective with the padded asynchronous one, and mismatched collectives deadlock.  The first call on a group is itself
    a (tiny, host-side) collective: every rank of the group must reach it, which all_gather_slabs_async guarantees."""
    key = id(group) if group is not None else 0
    if key not in _gather_sync_agreed:
        mine = _env_flag("EGS_GATHER_SYNC")
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            votes = [None] * dist.get_world_size(group)
            dist.all_gather_object(votes, mine, group=group)
            if len(set(votes)) != 1:
                raise RuntimeError("EGS_GATHER_SYNC differs between ranks (%s): set it identically on every rank" % votes)
        _gather_sync_agreed[key] = mine
    return _gather_sync_agreed[key]


def _host_staged(group=None):
    """gloo moves host memory: device slabs are staged through the host around the collective."""
    return dist.get_backend(group) == "gloo"


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
    device = local.device
    if local.is_cuda and _host_staged(group):
        local = local.cpu()
    out = torch.empty((world_size * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_complex():
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(local), group=group)
    else:
        dist.all_gather_into_tensor(out, local, group=group)
    return out[:n_total].to(device)


class _GatherHandle:
    """Result of all_gather_slabs_async: wait() makes the current stream wait for the collective and returns the full
    tensor.  Holds the input slab and the output until then (the collective reads / writes them on RCCL's stream)."""

    def __init__(self, out, n_total, device, work=None, keep=None):
        self._out, self._n, self._device, self._work, self._keep = out, n_total, device, work, keep

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._keep = None
        return self._out[:self._n].to(self._device)


def all_gather_slabs_async(local, n_total, group=None):
    """all_gather_slabs without blocking the launch stream: on RCCL the collective runs on its own stream (after the work
    already queued on the current one) and the caller goes on to launch the next batch's kernels; .wait() joins.  With
    host-staged (gloo) groups, or with EGS_GATHER_SYNC=1, the exchange is done at once and the handle is already complete."""
    rank, world_size = world()
    if world_size == 1:
        return _GatherHandle(local, local.shape[0], local.device)
    if (local.is_cuda and _host_staged(group)) or gather_sync(group):
        # EGS_GATHER_SYNC=1: the blocking collective as a fallback (the overlapped form has only been rehearsed on a
        # one-rank RCCL group and on gloo: no multi-GPU node in reach of this build)
        return _GatherHandle(all_gather_slabs(local, n_total, group), n_total, local.device)
    return _gather_async(local, n_total, world_size, group)


def _gather_async(local, n_total, world_size, group=None):
    """The collective itself (separate so that a one-rank RCCL group can exercise it on a one-GPU box)."""
    per = (n_total + world_size - 1) // world_size
    pad = per - local.shape[0]
    if pad > 0:
        filler = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, filler], dim=0)
    local = local.contiguous()
    out = torch.empty((world_size * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_complex():
        work = dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(local), group=group, async_op=True)
    else:
        work = dist.all_gather_into_tensor(out, local, group=group, async_op=True)
    return _GatherHandle(out, n_total, local.device, work, local)


def all_reduce_slab_sum(full, group=None):
    """The histogram-only exchange of SURVEY.md 8(e): every rank has filled ITS batch rows of `full` (zeros elsewhere),
    one all_reduce(sum) completes it on every rank.  Bit-exact: each element receives exactly one non-zero term."""
    rank, world_size = world()
    if world_size == 1:
        return full
    device = full.device
    buf = full.cpu() if (full.is_cuda and _host_staged(group)) else full
    buf = buf.contiguous()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf.to(device)
