"""Multi-GPU layer of the memory path: one process per GPU, `torch.distributed` ("nccl" = RCCL over xGMI).

The recurrence couples the chunks of ONE video (step t reads the FIFO written by steps < t,
MemoryController.py:125-127,152), so the unit that shards without changing results is the video: each rank runs
whole videos (the reference trains/evaluates the same way - batch 1 per GPU, data parallel).  The only data-path
exchange is the all-gather of every rank's final memory state `[M,P,D]` (the north-star's "RCCL all-gather of the
final memory state"); it is issued asynchronously so it overlaps the next video's kernels.

xGMI is point-to-point (7 links per GPU): the message here is M*P*D*2 bytes per rank (3.2 MB at M=8, 25.7 MB at
M=64) - one all_gather_into_tensor per video, no bucketing needed, latency-dominated at M=8.
"""
import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank); a plain single-process run returns (0, 1, 0) without a group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("MAVLM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous balanced shard of `n_items` independent videos: the first n % world ranks get one extra."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def all_gather_memory_state(state: torch.Tensor, out: Optional[torch.Tensor] = None, async_op: bool = False, group=None):
    """Gather every rank's final memory state.  state: [M,P,D] (contiguous).  Returns (gathered [W,M,P,D], work).
    With async_op=True the caller waits on `work` (or the stream) before reading `gathered`."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if out is None:
        out = torch.empty((world,) + tuple(state.shape), device=state.device, dtype=state.dtype)
    if world == 1:
        out[0].copy_(state)
        return out, None
    work = dist.all_gather_into_tensor(out.view(-1), state.contiguous().view(-1), group=group, async_op=async_op)
    return out, work
