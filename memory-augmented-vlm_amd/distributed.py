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
    of memory tokens [g*M/W, (g+1)*M/W) and runs the FUSED engine on them (`mavlm_config.q_tokens`: one `mavlm_step` per
    chunk - the same launch sequence, kernels and device-resident K/V ring as the single-GPU path; round 3).  Per step:
      * the chunk's K/V projection and the K/V projection of the newest (gathered) memory are computed on every rank
        (redundant: 6 % + 3 % of a step's flops at M = 64, no communication),
      * evolution + formation run for the rank's own rows only (the small-grid plans of the kernels - split-KV attention,
        split-K GEMMs - keep the chip busy at 1/W of the rows); the last LayerNorm writes them into the rank's rows of the
        FIFO slot,
      * ONE all-gather, in place over the slot (M*P*D*2 bytes in total: 25.7 MB at M = 64), issued `async_op=True`, gives
        every rank the full entry for the next step's evolution; it overlaps the NEXT chunk's K/V projection
        (`step(seg, prefetch=next_seg)` -> `mavlm_project_chunk`), the one GEMM of a step that does not read the memory;
      * the per-frame attention scores are partial sums over the rank's query rows: all-reduced (F floats, fp32).
    Exact: same values as the single-GPU path up to fp32 summation order (the kernel plans depend on the row count).
    Inference only.  `projector` is a TransformerProjector whose parameters are replicated on every rank."""

    def __init__(self, projector, group=None):
        self.p = projector
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        M = projector.num_memory_tokens
        if M % self.world:
            raise ValueError(f"num_memory_tokens ({M}) must be a multiple of the group size ({self.world})")
        self.tokens = M // self.world
        self.token0 = self.rank * self.tokens
        self._engine = None
        self._pending = None
        self._n = 0
        self._packed_epoch = -1
        self._pre = None           # (the tensor given as `prefetch`, its contiguous form whose K/V sit in the workspace)

    # -- engine (a row shard of the projector's fused engine) ------------------------------------------------
    def _eng(self, device, dtype, frames):
        from .model.memory_module.MemoryController import _Engine
        p = self.p
        need = max(int(getattr(p.config, "max_chunk_frames", 32)), int(frames or 0))
        e = self._engine
        if e is None or e.device != device or e.dtype != dtype or e.c.max_chunk_frames < need:
            if e is not None and e.steps:
                raise RuntimeError("device / dtype / chunk size changed in the middle of a video")
            e = self._engine = _Engine(p, device, dtype, need, shard=(self.token0, self.tokens) if self.world > 1 else None)
        v = p._param_version()
        if e.version != v:
            fuser, temb = p._fuser_refs if p._fuser_refs is not None else (None, None)
            e.pack(p, fuser, temb)
            e.version = v
            self._packed_epoch = p._train_state["epoch"]
        return e

    def reset(self):
        from . import _capi as capi
        self.wait()
        self._n = 0
        self._pre = None
        if self._engine is not None:
            capi.check(capi.lib().mavlm_reset(self._engine.ctx), "mavlm_reset")
            if self.p._weights_maybe_stale(self._packed_epoch):
                self._engine.version = None
            self._engine.post_ln_probe()

    def wait(self):
        """the all-gather of the newest memory is complete (the compute stream is ordered behind it)"""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    @property
    def cache(self):
        e = self._engine
        if e is None:
            return []
        cap = e.c.cache_cap
        n = min(self._n, cap)
        return [e.mem_ring[(self._n - n + i) % cap] for i in range(n)]

    @torch.no_grad()
    def step(self, image_features: torch.Tensor, prefetch: Optional[torch.Tensor] = None):
        """One chunk [F,P,D] (already PE-added).  `prefetch`: the NEXT chunk (the tensor the next `step` will be given): its
        K/V projection is enqueued while the all-gather of this step's memory is in flight.
        Returns (memory_cache: list of full [M,P,D] memories, oldest first; scores [F])."""
        from . import _capi as capi
        from . import _ops as ops
        p = self.p
        F, P, D = image_features.shape
        # The prefetch cache of the C library matches on the POINTER of the chunk: when the chunk announced by the previous
        # step's `prefetch` had to be made contiguous, that copy - not a second one - is what this step must run on.
        pre, self._pre = self._pre, None
        if pre is not None and (image_features is pre[0] or (image_features.data_ptr() == pre[0].data_ptr() and
                                                               image_features.shape == pre[0].shape and
                                                               image_features.stride() == pre[0].stride())):
            x = pre[1]
        else:
            x = image_features.contiguous()
        e = self._eng(x.device, x.dtype, F)
        lib = capi.lib()
        self.wait()                                               # (normally already complete: see below)
        scores = torch.empty(F, device=x.device, dtype=torch.float32)
        capi.check(lib.mavlm_step(e.ctx, x.data_ptr(), F, scores.data_ptr(), 1, ops.stream_ptr()), "mavlm_step")
        self._n += 1
        slot = e.mem_ring[lib.mavlm_newest_slot(e.ctx)]            # [M,P,D]: this rank's rows are written, the others stale
        if self.world > 1:
            own = slot[self.token0:self.token0 + self.tokens]
            send = own if dist.get_backend(self.group) == "nccl" else own.clone()    # (in place over the slot: NCCL only)
            self._pending = dist.all_gather_into_tensor(slot.view(-1), send.reshape(-1), group=self.group, async_op=True)
            swork = dist.all_reduce(scores, group=self.group, async_op=True)
            if prefetch is not None:                              # overlaps the all-gather: needs nothing of the memory
                nx = prefetch.contiguous()
                capi.check(lib.mavlm_project_chunk(e.ctx, nx.data_ptr(), nx.shape[0], ops.stream_ptr()), "mavlm_project_chunk")
                self._pre = (prefetch, nx)                        # keeps the contiguous copy alive until the next step
            swork.wait()
            self.wait()
        return self.cache, scores.to(x.dtype)

    @torch.no_grad()
    def fused_memory(self, memory_fuser, type_embedding_row0: Optional[torch.Tensor] = None):
        """The Memory-Fuser MLP over the FIFO (llava_arch.py:545-554: `memory_fuser(cat(cache))` + token-type row 0), row
        sharded as the steps are: rank g fuses its memory tokens of every cached memory ([n, M/W, P, D] rows: the MLP is
        row-independent, two HIP GEMMs with the GELU in the first epilogue), ONE all-gather ([W, n, M/W, P, D]) completes
        the block on every rank.  Returns [n*M, P, D], oldest memory first - the rows the single-GPU path feeds the LLM."""
        from . import _capi as capi
        from . import _ops as ops
        self.wait()
        cache = self.cache
        n = len(cache)
        if n == 0:
            raise RuntimeError("fused_memory: no step since reset")
        M, P, D = cache[0].shape
        own = torch.stack([c[self.token0:self.token0 + self.tokens] for c in cache])          # [n, tokens, P, D]
        b2 = memory_fuser[2].bias.float()
        if type_embedding_row0 is not None:
            b2 = b2 + type_embedding_row0.float()
        h = ops.linear(own.view(-1, D), memory_fuser[0].weight, memory_fuser[0].bias.float(), capi.EPI_GELU)
        y = ops.linear(h, memory_fuser[2].weight, b2, capi.EPI_BIAS).view(n, self.tokens, P, D)
        if self.world == 1:
            return y.reshape(n * M, P, D)
        full = torch.empty((self.world, n, self.tokens, P, D), device=y.device, dtype=y.dtype)
        dist.all_gather_into_tensor(full.view(-1), y.reshape(-1), group=self.group)
        return full.permute(1, 0, 2, 3, 4).reshape(n * M, P, D)
