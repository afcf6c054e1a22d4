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


class RowShardedMemory:
    """ONE video's recurrence with the memory ROWS sharded over the ranks of a process group - SURVEY.md §8e option 2,
    BASELINE.json configs[3] ("a 1024-frame video ... chunks sharded across GPUs, RCCL all-gather of memory state").

    The chunks of a video cannot be sharded (step t reads the FIFO of steps < t), but inside a step every operation on
    the memory side is row-independent: q projection, attention rows, out dense, LayerNorm, MLP.  Rank g owns the rows
    of memory tokens [g*M/W, (g+1)*M/W); per step it
      * projects the chunk's K/V itself (redundant on every rank: 6 % of a step's flops at M = 64, no communication),
      * runs evolution + formation for its own rows only (the small-grid paths of the kernels - split-KV attention,
        split-K GEMMs - keep the chip busy at 1/W of the rows),
      * all-gathers the new memory rows (M*P*D*2 bytes in total: 25.7 MB at M = 64) so that every rank holds the full
        FIFO entry, and all-reduces the per-frame attention scores (F floats).
    Exact: same values as the single-GPU path up to fp32 summation order (the kernel plan depends on the row count).
    Inference only.  `projector` is a TransformerProjector whose parameters are replicated on every rank."""

    def __init__(self, projector, group=None):
        self.p = projector
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        M = projector.num_memory_tokens
        if M % self.world:
            raise ValueError(f"num_memory_tokens ({M}) must be a multiple of the group size ({self.world})")
        self.rows = (M // self.world) * projector.patch_size
        self.r0 = self.rank * self.rows
        self.reset()

    def reset(self):
        self.cache, self._kv, self._steps = [], [], 0

    @torch.no_grad()
    def step(self, image_features: torch.Tensor):
        """One chunk [F,P,D] (already PE-added).  Returns (memory_cache: list of full [M,P,D] memories, scores [F])."""
        from . import _autograd as ag
        from . import _ops as ops
        p = self.p
        F, P, D = image_features.shape
        M = p.num_memory_tokens
        R, dt = M * P, image_features.dtype
        cap = int(getattr(p.config, "cache_cap", 10))
        frames = image_features.reshape(F * P, D)
        sl = slice(self.r0, self.r0 + self.rows)
        if self.cache:
            evo = p.memory_update_attention
            while len(self._kv) < len(self.cache):                       # K/V of a cached memory: once, on every rank
                self._kv.append(ag.project_kv([evo], self.cache[len(self._kv)].reshape(R, D))[0])
            first = self._steps - len(self._kv)
            order = sorted(range(len(self._kv)), key=lambda i: (first + i) % cap)   # ring order, as the fused step
            k = torch.cat([self._kv[i][0] for i in order], dim=0)
            v = torch.cat([self._kv[i][1] for i in order], dim=0)
            m, _ = ag.attention_block(evo, self.cache[-1].reshape(R, D)[sl].contiguous(), k, v)
        else:
            m = (p.initial_memory + p.memory_pos_embed).to(dt).reshape(R, D)[sl].contiguous()
        atts = [layer.memory_segment_fusion_attention for layer in p.layers]
        kvs = ag.project_kv(atts, frames)
        stats = None
        for li, layer in enumerate(p.layers):
            a, stats = ag.attention_block(atts[li], m, kvs[li][0], kvs[li][1], want_stats=li == len(p.layers) - 1,
                                          patches_per_frame=P)
            m = ag.mlp_block(layer, a)
        full = torch.empty((R, D), device=m.device, dtype=dt)
        if self.world > 1:
            dist.all_gather_into_tensor(full.view(-1), m.contiguous().view(-1), group=self.group)
        else:
            full.copy_(m)
        q, kk, lse = stats
        att = atts[-1]
        hdw = 128 if att.attention_head_size <= 128 else att.attention_head_size
        part = ops.attention_colsum(q, kk, lse, att.num_attention_heads, head_dim=hdw,
                                    scale=ops.attn_scale(att.attention_head_size)).sum(dim=0)     # [S], local queries
        if self.world > 1:
            dist.all_reduce(part, group=self.group)
        scores = part.view(F, P).mean(dim=1).to(dt)
        self.cache.append(full.view(M, P, D))
        self._steps += 1
        if len(self.cache) > cap:
            drop = len(self.cache) - cap
            self.cache, self._kv = self.cache[drop:], self._kv[drop:]
        return self.cache, scores
