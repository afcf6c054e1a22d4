"""Multimodal glue of the memory path on the MI355X HIP engine.

Mirrors the memory-specific parts of the reference's llava/model/llava_arch.py so the HF/LLaVA backbone
(vision tower, mm_projector, Qwen2 LLM - all unchanged PyTorch-ROCm modules) keeps working:

  LlavaMetaModel            builds `recurrent_memory_transformer`, `memory_fuser`, `positional_encoding`,
                            `token_type_embedding` with the reference's names and hard-coded hyper-parameters
                            (llava_arch.py:117-150) - same state-dict keys, so checkpoints load unchanged.
  LlavaMetaForCausalLM      `prepare_inputs_labels_for_multimodal` (8 in / 6 out, llava_arch.py:388-878) for the
                            video path, `get_2dPool`, `encode_images`, `get_synced_dropout_decision`.
  video_memory_tokens       the per-video driver (llava_arch.py:502-557,613-629,705-731) as HIP launches only:
                            PE add -> chunk loop (mavlm_step) -> fuser MLP + type add + prompts/newlines concat
                            (mavlm_fuse_emit) written straight into one token block.

Non-video inputs behave as in the reference (round 3): a plain image batch takes the reference's tensor branch (:703,
backbone ops, restated in torch); a list without a video fails as the reference's memory loop does (IndexError); images
beside the video are dropped as the reference drops them.  The anyres / multi-patch merge modes (:630-697) are unreachable
in the reference's memory branch (its feature list only ever holds the memory and the fine frames) and are not restated.
"""
import math
import random
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import _capi as capi
from .. import _ops as ops
from .memory_module.MemoryController import BatchedProjector, Config, TransformerProjector
from .memory_module.position_encoding import TemporalPositionalEncoding
from .memory_module.segment import uniform_segment_variant

IGNORE_INDEX = -100          # llava/constants.py:7
IMAGE_TOKEN_INDEX = -200     # llava/constants.py:8

# Qwen2-tokenizer ids of the two fixed prompts (llava_arch.py:708,714):
# "This is a high-level summary of the video:" / "These are sampled visual frames from the video:"
MEMORY_PROMPT_IDS = (1986, 374, 264, 1550, 11591, 12126, 315, 279, 2766, 25)
FRAME_PROMPT_IDS = (9485, 525, 48876, 9124, 14087, 504, 279, 2766, 25)


class MemoryFuserMLP(nn.Sequential):
    """`memory_fuser` = Linear(D,4D) -> GELU(erf) -> Linear(4D,D) (llava_arch.py:132-136); state-dict keys
    `0.weight, 0.bias, 2.weight, 2.bias` as the reference's nn.Sequential.  forward = two MFMA GEMMs with the
    GELU fused into the first epilogue."""

    def __init__(self, hidden):
        super().__init__(nn.Linear(hidden, hidden * 4), nn.GELU(), nn.Linear(hidden * 4, hidden))

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .. import _autograd as ag            # training: same kernels as autograd Functions
            return ag.fuser_mlp(self, x)
        x2 = x.reshape(-1, x.shape[-1])
        u = ops.linear(x2, self[0].weight, self[0].bias.float(), capi.EPI_GELU)
        y = ops.linear(u, self[2].weight, self[2].bias.float(), capi.EPI_BIAS)
        return y.reshape(x.shape)


def sample_frame_count(num_frames: int) -> int:
    """F0 -> F  (llava_arch.py:437-445): all frames below 32, else a multiple of 32 and at least 64."""
    if num_frames < 32:
        return num_frames
    return max(64, (num_frames // 32) * 32)


def sample_frame_indices(num_frames: int) -> torch.Tensor:
    """llava_arch.py:451 - same call the reference makes (CPU float32 linspace, truncation)."""
    return torch.linspace(0, num_frames - 1, steps=sample_frame_count(num_frames)).long()


def fine_frame_indices(num_frames: int, want: int = 32) -> torch.Tensor:
    """llava_arch.py:513-522 - rounded, clamped linspace over the sampled frames."""
    n = min(want, num_frames)
    idx = torch.round(torch.linspace(0, num_frames - 1, steps=n)).long()
    return torch.clamp(idx, 0, num_frames - 1)


class LlavaMetaModel:
    """Mixin for the inner model (the class that owns `embed_tokens`).  Vision modules are the backbone's: attach
    `vision_tower`, `mm_projector`, `image_newline` as the host model does."""

    def __init__(self, config):
        super(LlavaMetaModel, self).__init__(config)
        hidden = getattr(config, "hidden_size", 896)
        c = Config()
        c.mm_hidden_size = hidden
        c.mm_hidden_act = "relu"
        c.mm_num_attention_heads = 8
        c.patch_size = 196
        c.mm_attention_probs_dropout_prob = 0.1
        c.mm_layer_norm_eps = 1e-12
        c.mm_hidden_dropout_prob = 0.1
        c.mm_intermediate_size = 4 * hidden
        c.num_memory_tokens = getattr(config, "num_memory_tokens", 8)     # reference constant: 8
        c.depth = 2
        c.mm_dtype = torch.float16
        c.cache_cap = getattr(config, "memory_cache_cap", 10)
        self.recurrent_memory_transformer = TransformerProjector(c)
        self.memory_fuser = MemoryFuserMLP(hidden)
        self.positional_encoding = TemporalPositionalEncoding(
            max_frames=getattr(config, "memory_max_frames", 600), embed_dim=hidden, learnable=False)
        self.token_type_embedding = nn.Embedding(2, hidden)
        self.recurrent_memory_transformer.bind_fuser(self.memory_fuser, self.token_type_embedding)

    def get_vision_tower(self):
        vt = getattr(self, "vision_tower", None)
        return vt[0] if type(vt) is list else vt


_INDEX_CACHE = {}


def _device_indices(idx_cpu: torch.Tensor, device) -> torch.Tensor:
    """int64 index vectors are tiny and repeat across videos of the same length: cache the device copy so the
    steady state has no H2D copy (a pageable copy would wait for the stream to drain)."""
    key = (str(device), idx_cpu.numel(), idx_cpu.numpy().tobytes())
    t = _INDEX_CACHE.get(key)
    if t is None:
        if len(_INDEX_CACHE) > 256:
            _INDEX_CACHE.clear()
        t = idx_cpu.to(device=device, dtype=torch.int64)
        _INDEX_CACHE[key] = t
    return t


def video_token_rows(num_frames: int, mem_tokens: int, patches: int = 196, with_frames: bool = True, chunk: int = 32,
                     fine_frames: int = 32, cache_cap: int = 10, n_mem_prompt: int = len(MEMORY_PROMPT_IDS),
                     n_frame_prompt: int = len(FRAME_PROMPT_IDS)) -> int:
    """Rows of the token block video_memory_tokens() produces for a T-frame video (host arithmetic only), so a caller
    can allocate the final sequence once and have the block emitted in place."""
    n = min(-(-num_frames // chunk), cache_cap)
    rows = n_mem_prompt + n * mem_tokens * patches + 1
    if with_frames:
        rows += n_frame_prompt + min(fine_frames, num_frames) * patches + 1
    return rows


PROJECT_AHEAD = None      # None: automatic (TransformerProjector.ahead_ok), True: wherever the memory rows are few, False: never


def video_memory_tokens(model, image: torch.Tensor, frame_idx_cpu: torch.Tensor, memory_prompt_embeds: torch.Tensor,
                        frame_prompt_embeds: torch.Tensor, image_newline: torch.Tensor, with_frames: bool = True,
                        chunk: int = 32, fine_frames: int = 32, out: Optional[torch.Tensor] = None):
    """Per-video memory path.  `model` owns the four memory sub-modules; `image` = pooled frame tokens [T,196,D]
    on the GPU; `frame_idx_cpu` = original frame indices [T] (host).  Returns (tokens [rows,D], info dict).

    tokens = [mem_prompt ; fused memory (oldest first) ; newline ; frame_prompt ; fine frames ; newline]
    (llava_arch.py:620-629,729-731); with_frames=False reproduces the frame-dropout branch (:720-725).
    `out`: optional contiguous [rows, D] destination (e.g. a slice of the final inputs_embeds buffer)."""
    if not image.is_cuda:
        raise capi.MavlmError("video_memory_tokens: frame tokens are not on a GPU (no CPU fallback)")
    pe: TemporalPositionalEncoding = model.positional_encoding
    rm: TransformerProjector = model.recurrent_memory_transformer
    T, P, D = image.shape
    pe.check_indices(frame_idx_cpu)                                                       # ValueError, :73-76
    x = pe(image, _device_indices(frame_idx_cpu, image.device), indices_checked=True)     # :510-511
    fine_cpu = fine_frame_indices(T, fine_frames)                                         # :513-522
    bounds = uniform_segment_variant(T, chunk)                                            # :528
    rm.memory_cache = []                                                                  # :532
    # (few memory tokens: the next chunk's K/V projection - the largest GEMM of a step there - runs on a side stream beside this
    #  chunk's small-grid kernels, TransformerProjector.project_ahead; PROJECT_AHEAD = False switches it off)
    ahead = PROJECT_AHEAD is not False and len(bounds) > 2 and rm.ahead_ok(force=PROJECT_AHEAD is True) and \
        max(bounds[i + 1] - bounds[i] for i in range(len(bounds) - 1)) <= int(getattr(rm.config, "max_chunk_frames", 32))
    for i in range(len(bounds) - 1):                                                      # :534-537
        if ahead and i + 2 < len(bounds):
            rm.project_ahead(x[bounds[i + 1]:bounds[i + 2]])
        rm(x[bounds[i]:bounds[i + 1]])
    if rm._cache_mode == "autograd" or _tail_wants_grad(model, memory_prompt_embeds, frame_prompt_embeds, image_newline):
        # training.  With `recurrent_memory_transformer` frozen but a trainable fuser / token-type embedding / newline /
        # prompt embedding (mm_tunable_parts="larimar_model", train.py:1708-1713) the chunks above ran on the engine
        # (nothing to differentiate there) and only the tail below records a graph - on CLONES of the ring views, which
        # the next video would overwrite before backward.
        return _video_memory_tokens_autograd(model, rm, x, fine_cpu, memory_prompt_embeds, frame_prompt_embeds,
                                             image_newline, with_frames, out)
    eng = rm.engine(image.device, image.dtype)
    n = len(rm.memory_cache)
    R = rm.num_memory_tokens * P
    n_fine = fine_cpu.numel()
    rows = memory_prompt_embeds.shape[0] + n * R + 1
    if with_frames:
        rows += frame_prompt_embeds.shape[0] + n_fine * P + 1
    if out is None:
        out = torch.empty((rows, D), device=image.device, dtype=image.dtype)
    elif tuple(out.shape) != (rows, D) or out.dtype != image.dtype or not out.is_contiguous() or not out.is_cuda:
        raise capi.MavlmError(f"video_memory_tokens: `out` must be a contiguous [{rows},{D}] {image.dtype} GPU tensor")
    mp = memory_prompt_embeds.to(image.dtype).contiguous()
    fp = frame_prompt_embeds.to(image.dtype).contiguous()
    nl = image_newline.to(device=image.device, dtype=image.dtype).contiguous()
    import ctypes
    written = ctypes.c_int64(0)
    capi.check(capi.lib().mavlm_fuse_emit(eng.ctx, x.data_ptr(), _device_indices(fine_cpu, image.device).data_ptr(), n_fine,
                                          mp.data_ptr(), mp.shape[0], fp.data_ptr(), fp.shape[0], nl.data_ptr(),
                                          1 if with_frames else 0, out.data_ptr(), rows, ctypes.byref(written),
                                          ops.stream_ptr()), "mavlm_fuse_emit")
    assert written.value == rows
    info = {"num_memories": n, "pe_frames": x, "fine_idx": fine_cpu, "memory_rows": (mp.shape[0], mp.shape[0] + n * R)}
    return out, info


@torch.no_grad()
def video_memory_tokens_batched(model, bp: BatchedProjector, images, frame_idx_cpu: torch.Tensor,
                                memory_prompt_embeds: torch.Tensor, frame_prompt_embeds: torch.Tensor,
                                image_newline: torch.Tensor, with_frames: bool = True, chunk: int = 32,
                                fine_frames: int = 32, out: Optional[torch.Tensor] = None):
    """`video_memory_tokens` for B videos of the SAME length stepped together (row batch, `BatchedProjector`): every
    weight-shared GEMM / LayerNorm of the path runs once over the stacked memory rows of all videos.  `images`: B tensors
    [T,196,D]; `frame_idx_cpu`: the original frame indices (shared: same length, same sampling).  Returns
    (tokens [B, rows, D] - video b's block is tokens[b] -, info).  Inference only."""
    B = bp.batch
    if len(images) != B:
        raise capi.MavlmError(f"video_memory_tokens_batched: {B} videos expected")
    T, P, D = images[0].shape
    for im in images:
        if not im.is_cuda or tuple(im.shape) != (T, P, D) or im.dtype != images[0].dtype:
            raise capi.MavlmError("video_memory_tokens_batched: GPU frame tokens of one shape / dtype expected")
    pe: TemporalPositionalEncoding = model.positional_encoding
    rm: TransformerProjector = model.recurrent_memory_transformer
    dev, dt = images[0].device, images[0].dtype
    pe.check_indices(frame_idx_cpu)
    idx_dev = _device_indices(frame_idx_cpu, dev)
    xs = [pe(im, idx_dev, indices_checked=True) for im in images]                           # :510-511
    fine_cpu = fine_frame_indices(T, fine_frames)                                          # :513-522
    bounds = uniform_segment_variant(T, chunk)                                             # :528
    bp.reset()                                                                             # :532
    for i in range(len(bounds) - 1):                                                       # :534-537
        bp.step([x[bounds[i]:bounds[i + 1]] for x in xs])
    eng = bp.engine(dev, dt)
    n = min(len(bounds) - 1, eng.c.cache_cap)
    R = rm.num_memory_tokens * P
    n_fine = fine_cpu.numel()
    rows = memory_prompt_embeds.shape[0] + n * R + 1
    if with_frames:
        rows += frame_prompt_embeds.shape[0] + n_fine * P + 1
    if out is None:
        out = torch.empty((B, rows, D), device=dev, dtype=dt)
    elif tuple(out.shape) != (B, rows, D) or out.dtype != dt or not out.is_contiguous() or not out.is_cuda:
        raise capi.MavlmError(f"video_memory_tokens_batched: `out` must be a contiguous [{B},{rows},{D}] {dt} GPU tensor")
    mp = memory_prompt_embeds.to(dt).contiguous()
    fp = frame_prompt_embeds.to(dt).contiguous()
    nl = image_newline.to(device=dev, dtype=dt).contiguous()
    import ctypes
    written = ctypes.c_int64(0)
    xptrs = (capi.vp * B)(*[x.data_ptr() for x in xs])
    capi.check(capi.lib().mavlm_fuse_emit_batch(eng.ctx, xptrs, _device_indices(fine_cpu, dev).data_ptr(), n_fine,
                                                mp.data_ptr(), mp.shape[0], fp.data_ptr(), fp.shape[0], nl.data_ptr(),
                                                1 if with_frames else 0, out.data_ptr(), rows, ctypes.byref(written),
                                                ops.stream_ptr()), "mavlm_fuse_emit_batch")
    assert written.value == rows
    info = {"num_memories": n, "pe_frames": xs, "fine_idx": fine_cpu, "memory_rows": (mp.shape[0], mp.shape[0] + n * R),
            "frame_scores": bp.frame_scores}
    return out, info


def _tail_wants_grad(model, *tensors) -> bool:
    """Can anything AFTER the recurrent steps receive a gradient?  (`memory_fuser`, `token_type_embedding`, and the
    tensors the caller hands in: prompt embeddings of a trainable `embed_tokens`, `image_newline`.)"""
    if not torch.is_grad_enabled():
        return False
    mods = (getattr(model, "memory_fuser", None), getattr(model, "token_type_embedding", None))
    return any(p.requires_grad for m in mods if m is not None for p in m.parameters()) or \
        any(t is not None and t.requires_grad for t in tensors)


def path_wants_grad(model, *tensors) -> bool:
    """Any gradient consumer on the memory path: the recurrent transformer or the tail (see _tail_wants_grad)."""
    return torch.is_grad_enabled() and (any(p.requires_grad for p in model.recurrent_memory_transformer.parameters())
                                        or _tail_wants_grad(model, *tensors))


def _video_memory_tokens_autograd(model, rm, x, fine_cpu, memory_prompt_embeds, frame_prompt_embeds, image_newline,
                                  with_frames, out):
    """Training-mode tail of video_memory_tokens (llava_arch.py:545-554,620-629): the fuser MLP runs as HIP autograd
    Functions; the type-embedding adds and the concatenation are torch ops so that autograd routes the gradients of
    token_type_embedding, the prompt embeddings and image_newline exactly as in the reference."""
    from .. import _autograd as ag
    P, D = x.shape[1], x.shape[2]
    dt = x.dtype
    temb = model.token_type_embedding.weight
    # engine-mode cache entries are ring views: torch.cat copies them (the graph must not alias the ring)
    mem = torch.cat(rm.memory_cache, dim=0)                                               # :545
    fused = ag.fuser_mlp(model.memory_fuser, mem, temb[0]).reshape(-1, D)                  # :546-553
    nl = image_newline.to(device=x.device, dtype=dt).reshape(1, D)
    parts = [memory_prompt_embeds.to(dt), fused, nl]
    n = len(rm.memory_cache)
    R = rm.num_memory_tokens * P
    if with_frames:
        fine = x[_device_indices(fine_cpu, x.device)] + temb[1].to(dt)                     # :513-524,554
        parts += [frame_prompt_embeds.to(dt), fine.reshape(-1, D), nl]
    tokens = torch.cat(parts, dim=0)
    if out is not None:
        raise capi.MavlmError("video_memory_tokens: `out=` is an inference-path feature (no autograd through a "
                              "caller-owned buffer)")
    mp_rows = memory_prompt_embeds.shape[0]
    info = {"num_memories": n, "pe_frames": x, "fine_idx": fine_cpu, "memory_rows": (mp_rows, mp_rows + n * R)}
    return tokens, info


_SIDE_STREAMS = {}


def _side_streams(device_index: int, n: int):
    """The first n side streams of a device, shared by every pool of the process.  HIP multiplexes its streams onto a few
    hardware queues (4 by default): a second pool with streams of its own ended up sharing queues and ran its two videos
    one after the other (measured at M = 8: 2.11 instead of 1.50 ms per pair of videos after another pool had been used)."""
    have = _SIDE_STREAMS.setdefault(device_index, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device_index))
    return have[:n]


class MemoryPathPool:
    """Keeps `n` videos in flight on `n` HIP streams over ONE set of weights.

    Why: at the reference shapes every kernel of the path launches a grid that is not a multiple of the 256 CUs
    (M*196 rows -> 1.5 / 3.06 "rounds"), so ~23 % of each kernel runs on a partly idle chip.  Kernels of an
    independent video on another stream fill those tails (measured +21 % frames/s with 2 videos in flight).
    Videos are independent units (the recurrence is per video), so results are bit-identical to the serial path.

    usage:  pool = MemoryPathPool(model, 2)
            outs = pool.run([(frames0, idx0), (frames1, idx1), ...], mem_prompt, frame_prompt, newline)

    `batch` > 1 (round 3): each stream steps `batch` videos TOGETHER as a row batch (`BatchedProjector`: the memory rows of
    the videos stacked into every weight-shared GEMM / LayerNorm launch) instead of one.  Consecutive videos of equal length
    and equal frame indices form a batch; whatever does not fill one runs through the single-video slots.  At the
    reference's 8 memory tokens this is what fills the chip (1568-row operands otherwise).
    """

    def __init__(self, model, n: int = 2, batch: int = 1):
        rm = model.recurrent_memory_transformer
        self.model = model
        # More than LN_MAX_STREAMS streams: the forward-progress argument of the fused dense + residual + LayerNorm kernel
        # (waiting workgroups per XCD, include/mavlm.h MAVLM_LN_MAX_STREAMS) no longer covers the pool - every slot of such
        # a pool runs the two-kernel form (GEMM + row LayerNorm; same fp32 inputs, statistics added in another order), on
        # recurrent states of its own (the model's own engine keeps the fused form for serial use).
        self.fused_ln_never = n > capi.LN_MAX_STREAMS
        if self.fused_ln_never:
            self.slots = [_ReplicaView(model, rm.spawn_replica(fused_ln_never=True)) for _ in range(n)]
        else:
            self.slots = [model] + [_ReplicaView(model, rm.spawn_replica()) for _ in range(n - 1)]
        self.batch = int(batch)
        self.bslots = [BatchedProjector(rm, self.batch, fused_ln_never=self.fused_ln_never) for _ in range(n)] if self.batch > 1 else []
        self.streams = None

    @torch.no_grad()           # inference feature: the replicas' FIFOs are ring views, not autograd tensors
    def run(self, videos, memory_prompt_embeds, frame_prompt_embeds, image_newline, with_frames: bool = True):
        if self.streams is None:
            self.streams = _side_streams(torch.cuda.current_device(), len(self.slots))
        cur = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(cur)
        outs = [None] * len(videos)
        singles = list(range(len(videos)))
        if self.batch > 1:
            # groups of `batch` consecutive videos with the same shape and frame indices step together
            singles, i, g = [], 0, 0
            while i < len(videos):
                grp = videos[i:i + self.batch]
                same = len(grp) == self.batch and all(v[0].shape == grp[0][0].shape and torch.equal(v[1], grp[0][1])
                                                      for v in grp[1:])
                if not same:
                    singles.append(i)
                    i += 1
                    continue
                k = g % len(self.bslots)
                with torch.cuda.stream(self.streams[k]):
                    toks = video_memory_tokens_batched(self.model, self.bslots[k], [v[0] for v in grp], grp[0][1],
                                                       memory_prompt_embeds, frame_prompt_embeds, image_newline, with_frames)[0]
                    for j, v in enumerate(grp):
                        outs[i + j] = toks[j]
                        v[0].record_stream(self.streams[k])
                i += self.batch
                g += 1
        for j, i in enumerate(singles):
            frames, idx_cpu = videos[i]
            k = j % len(self.slots)
            with torch.cuda.stream(self.streams[k]):
                outs[i] = video_memory_tokens(self.slots[k], frames, idx_cpu, memory_prompt_embeds, frame_prompt_embeds,
                                              image_newline, with_frames)[0]
                frames.record_stream(self.streams[k])
        for st in self.streams:
            cur.wait_stream(st)
        return outs


class GraphedVideoMemory:
    """hipGraph-captured memory path for one video shape (T frames): PE add, every chunk step (evolution +
    formation, FIFO slots baked in), fuser MLP + type add + concat - ONE graph launch per video instead of ~45 kernel
    launches (BASELINE.json configs[2]: "hipGraph-captured per-chunk memory update").  Possible because no entry
    point of the C ABI allocates or synchronises and the recurrent state lives in caller-owned buffers.

    usage:  g = GraphedVideoMemory(model, T=256, frame_idx_cpu=idx)        # captures on the current device
            tokens = g(frames, mem_prompt, frame_prompt, newline)          # copies into the static inputs, replays
    The returned tensor is the graph's static output buffer (valid until the next call).  A new shape (T, indices,
    with_frames) needs a new instance."""

    @torch.no_grad()
    def __init__(self, model, T: int, frame_idx_cpu: torch.Tensor, with_frames: bool = True, slot=None):
        rm = model.recurrent_memory_transformer
        self.view = slot if slot is not None else _ReplicaView(model, rm.spawn_replica())
        self.idx = frame_idx_cpu.clone()
        self.with_frames = with_frames
        p = next(rm.parameters())
        D, P = rm.hidden_size, rm.patch_size
        dev, dt = p.device, p.dtype
        self.x = torch.zeros((T, P, D), device=dev, dtype=dt)
        self.mp = torch.zeros((len(MEMORY_PROMPT_IDS), D), device=dev, dtype=dt)
        self.fp = torch.zeros((len(FRAME_PROMPT_IDS), D), device=dev, dtype=dt)
        self.nl = torch.zeros((D,), device=dev, dtype=dt)
        # warm-up outside capture: engine creation, weight packing, kernel attribute calls, index uploads
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            video_memory_tokens(self.view, self.x, self.idx, self.mp, self.fp, self.nl, with_frames)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out, _ = video_memory_tokens(self.view, self.x, self.idx, self.mp, self.fp, self.nl, with_frames)
        # what the module's host-side state looks like after this video (ring views of the FIFO, scores of the chunks): a replay
        # only re-runs the device work, so __call__ puts these back - `recurrent_memory_transformer.memory_cache` read after a
        # replayed forward is the replayed video's cache, as after an eager forward (matters when the slot is the model itself)
        rm_ = self.view.recurrent_memory_transformer
        self._cache = list(rm_.memory_cache)

    @torch.no_grad()
    def __call__(self, frames, memory_prompt_embeds, frame_prompt_embeds, image_newline):
        self.x.copy_(frames)
        self.mp.copy_(memory_prompt_embeds)
        self.fp.copy_(frame_prompt_embeds)
        self.nl.copy_(image_newline)
        self.graph.replay()
        self.view.recurrent_memory_transformer._memory_cache = list(self._cache)
        return self.out


class _ReplicaView:
    """The four memory sub-modules as seen by one pool slot (shared weights, private recurrent state)."""

    def __init__(self, model, projector):
        self.positional_encoding = model.positional_encoding
        self.memory_fuser = model.memory_fuser
        self.token_type_embedding = model.token_type_embedding
        self.recurrent_memory_transformer = projector


class LlavaMetaForCausalLM:
    """Mixin for the *ForCausalLM wrapper: needs `get_model()`, `.config`, `.device`."""

    def get_model(self):
        raise NotImplementedError

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    def get_2dPool(self, image_feature, stride=2):
        """[F, side*side, D] -> [F, ceil(side/stride)^2, D]  (llava_arch.py:277-297).  Step before the path
        (SURVEY.md §8f rank 1).  The bilinear mode the reference scripts use runs as a HIP kernel on 16-bit GPU
        tensors (no NCHW permute round trip); the average / max modes stay backbone ops."""
        side = self.get_vision_tower().num_patches_per_side
        nf, _, nd = image_feature.shape
        mode = self.config.mm_spatial_pool_mode
        if mode == "bilinear" and image_feature.is_cuda and image_feature.dtype in (torch.bfloat16, torch.float16) \
                and nd % 8 == 0:
            return ops.pool_bilinear(image_feature.contiguous(), side, stride)      # HIP kernel (channels stay last)
        x = image_feature.view(nf, side, side, -1).permute(0, 3, 1, 2).contiguous()
        if mode == "average":
            x = nn.functional.avg_pool2d(x, stride)
        elif mode == "max":
            x = nn.functional.max_pool2d(x, stride)
        elif mode == "bilinear":
            size = [math.ceil(side / stride), math.ceil(side / stride)]
            x = nn.functional.interpolate(x, size=size, mode="bilinear")
        else:
            raise ValueError(f"Unexpected mm_spatial_pool_mode: {mode}")
        return x.permute(0, 2, 3, 1).reshape(nf, -1, nd)

    def encode_images(self, images):
        feats = self.get_model().get_vision_tower()(images)
        return self.get_model().mm_projector(feats).detach()      # llava_arch.py:299-304

    def get_synced_dropout_decision(self, prob: float = 0.5):
        """Shared Bernoulli(prob) across ranks (llava_arch.py:378-386): rank-0 draw, 1-element broadcast."""
        if not dist.is_initialized():
            return torch.rand(1).item() < prob
        flag = torch.zeros(1, device=self.device)
        if dist.get_rank() == 0:
            flag.fill_(1.0 if torch.rand(1).item() < prob else 0.0)
        dist.broadcast(flag, src=0)
        return bool(flag.item())

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, modalities=["image"], image_sizes=None):
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels        # :392-394
        if isinstance(modalities, str):
            modalities = [modalities]
        model = self.get_model()
        if not (type(images) is list or images.ndim == 5):
            # A plain [N,3,H,W] image batch (llava_arch.py:703): the reference encodes it and then - its prompt insertion is
            # unconditional (:705-731) - uses the features of image 0 as "memory" and of image 1 as "frames" (an IndexError
            # for a single image).  Backbone ops only; restated as it behaves.
            feats = self.encode_images(images)                                                       # :703
            dev = feats.device
            mem_prompt = model.embed_tokens(torch.tensor([MEMORY_PROMPT_IDS], device=dev)).squeeze(0)
            frame_prompt = model.embed_tokens(torch.tensor([FRAME_PROMPT_IDS], device=dev)).squeeze(0)
            drop = self.get_synced_dropout_decision(prob=0.5) and bool(getattr(self, "training", False)) and \
                bool(getattr(self.config, "dropout_frames", False))                                  # :719-720
            if drop:
                tokens = torch.cat((mem_prompt, feats[0]), dim=0)                                    # :722-725
            else:
                tokens = torch.cat((mem_prompt, feats[0], frame_prompt, feats[1]), dim=0)            # :729-731
            return splice_into_text(self, model, [tokens], input_ids, position_ids, attention_mask, past_key_values, labels)
        images = [x.unsqueeze(0) if x.ndim == 3 else x for x in images] if type(images) is list else list(images)
        vids = [i for i in range(min(len(modalities), len(images))) if modalities[i] == "video"]     # :403-406
        if not vids:
            # the reference: every non-video entry is skipped by the memory loop (:487-490), `image_features` ends up empty
            # and the prompt insertion indexes it (:722 / :730)
            raise IndexError("list index out of range (no video among `images`: the reference's memory branch skips "
                             "non-video entries, llava_arch.py:487-490, and then indexes an empty feature list, :730)")
        if len(vids) != 1:
            raise NotImplementedError("the memory path supports one video per forward (llava_arch.py:436: 'Now support "
                                      "only batch size of 1'; a second video fails the reference's patch-grid assert, :636)")
        # non-video entries beside the video are encoded and then DROPPED by the reference (:487-490, 556): dropped here too
        images = [images[vids[0]]]
        if getattr(self.config, "mm_newline_position", "one_token") != "one_token" or \
                "unpad" not in getattr(self.config, "mm_patch_merge_type", "flat"):
            raise NotImplementedError("memory path: mm_newline_position='one_token' with an *_unpad merge type only")
        video = images[0]
        idx_cpu = sample_frame_indices(video.shape[0])                                          # :437-451
        feats = self.encode_images(video[idx_cpu.to(video.device)])                             # :457-481
        pooled = self.get_2dPool(feats)                                                         # :495
        dev = pooled.device
        mem_prompt = model.embed_tokens(torch.tensor([MEMORY_PROMPT_IDS], device=dev)).squeeze(0)   # :708-709
        frame_prompt = model.embed_tokens(torch.tensor([FRAME_PROMPT_IDS], device=dev)).squeeze(0)  # :714-715
        training = bool(getattr(self, "training", False))
        dropout_frames = getattr(self.config, "dropout_frames", False)
        # :719 - called on EVERY forward, inference included, exactly as the reference: without a process group one local
        # draw; with one, rank 0 draws and every rank joins the 1-element broadcast (ranks of the reference and of this
        # build can therefore be mixed in one job, and the RNG streams of the non-zero ranks stay the reference's)
        drop = self.get_synced_dropout_decision(prob=0.5) and training and bool(dropout_frames)   # :720
        # Step after the path.  Fast path = what the reference's memory branch supports anyway
        # (batch 1, one image placeholder, llava_arch.py:436): the final [1, L, D] buffer is allocated once and the
        # HIP path writes the video block straight into it; everything else goes through the general splice.
        direct = self._direct_emit(model, pooled, idx_cpu, mem_prompt, frame_prompt, not drop, input_ids, position_ids,
                                   attention_mask, past_key_values, labels)
        if direct is not None:
            return direct
        tokens = self._video_tokens(model, pooled, idx_cpu, mem_prompt, frame_prompt, not drop)
        return splice_into_text(self, model, [tokens], input_ids, position_ids, attention_mask, past_key_values, labels)

    # -- hipGraph replay for repeated video shapes (round 4) -------------------------------------------------------------
    def enable_memory_graphs(self, capacity: int = 4):
        """Inference: serve videos of a shape seen before ((frames, frame indices, with_frames) of the sampled video) by replaying
        a captured hipGraph of the whole per-video launch sequence (`GraphedVideoMemory` on the model's own engine: PE add,
        every chunk step, fuser + emit) instead of ~45 launches; the first occurrence of a shape runs eagerly, the second
        captures.  Bit-identical to the eager path (`test_graph_capture_replay_bit_identical`).  `capacity` graphs are kept
        (least recently used out); 0 switches the cache off.  Each graph holds static copies of its input frames and of its
        token block."""
        self._mem_graph_capacity = int(capacity)
        self._mem_graphs = {}
        self._mem_graph_seen = {}

    def _video_tokens(self, model, pooled, idx_cpu, mem_prompt, frame_prompt, with_frames, out=None):
        """video_memory_tokens, through the graph cache when it applies (inference, shape seen before)"""
        cap = getattr(self, "_mem_graph_capacity", 0)
        if (cap <= 0 or path_wants_grad(model, mem_prompt, frame_prompt, getattr(model, "image_newline", None))
                or torch.cuda.is_current_stream_capturing()):
            return video_memory_tokens(model, pooled, idx_cpu, mem_prompt, frame_prompt, model.image_newline, with_frames, out=out)[0]
        key = (pooled.shape[0], tuple(int(i) for i in idx_cpu.tolist()), bool(with_frames), pooled.dtype, pooled.device.index)
        g = self._mem_graphs.pop(key, None)
        if g is None:
            n = self._mem_graph_seen.get(key, 0) + 1
            self._mem_graph_seen[key] = n
            if len(self._mem_graph_seen) > 64:
                self._mem_graph_seen.pop(next(iter(self._mem_graph_seen)))
            if n < 2:
                return video_memory_tokens(model, pooled, idx_cpu, mem_prompt, frame_prompt, model.image_newline, with_frames,
                                           out=out)[0]
            with torch.no_grad():
                g = GraphedVideoMemory(model, pooled.shape[0], idx_cpu, with_frames, slot=model)
            while len(self._mem_graphs) >= cap:
                self._mem_graphs.pop(next(iter(self._mem_graphs)))
        self._mem_graphs[key] = g                          # (re-inserted last: most recently used)
        tokens = g(pooled, mem_prompt, frame_prompt, model.image_newline)
        if out is not None:
            out.copy_(tokens)
            return out
        return tokens.clone()                              # (the graph's static buffer is overwritten by its next replay)

    def _direct_emit(self, model, pooled, idx_cpu, mem_prompt, frame_prompt, with_frames, input_ids, position_ids,
                     attention_mask, past_key_values, labels):
        if input_ids.shape[0] != 1:
            return None
        if path_wants_grad(model, mem_prompt, frame_prompt, getattr(model, "image_newline", None)):
            return None                                    # training: tokens must stay in the autograd graph
        mask = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
        ids = input_ids[0][mask[0]]
        pos = torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist()
        if len(pos) != 1:
            return None
        p = pos[0]
        rm = model.recurrent_memory_transformer
        rows = video_token_rows(pooled.shape[0], rm.num_memory_tokens, rm.patch_size, with_frames,
                                cache_cap=getattr(rm.config, "cache_cap", 10))
        n_text = ids.shape[0] - 1
        total = n_text + rows
        max_tok = getattr(self.config, "tokenizer_model_max_length", None)
        if max_tok is not None and total > max_tok:
            return None                                    # truncation cuts into the block: general path
        text = model.embed_tokens(torch.cat([ids[:p], ids[p + 1:]]))
        emb = torch.empty((1, total, pooled.shape[-1]), device=pooled.device, dtype=pooled.dtype)
        emb[0, :p] = text[:p].to(emb.dtype)
        emb[0, p + rows:] = text[p:].to(emb.dtype)
        self._video_tokens(model, pooled, idx_cpu, mem_prompt, frame_prompt, with_frames, out=emb[0, p:p + rows])
        if text.dtype != emb.dtype:
            emb = emb.to(text.dtype)
        out_labels = None
        if labels is not None:
            lab = labels[0][mask[0]]
            out_labels = torch.cat([lab[:p], torch.full((rows,), IGNORE_INDEX, device=lab.device, dtype=lab.dtype),
                                    lab[p + 1:]])[None]
        out_mask = None if attention_mask is None else torch.ones((1, total), device=attention_mask.device,
                                                                  dtype=attention_mask.dtype)
        out_pos = None if position_ids is None else torch.arange(total, device=position_ids.device,
                                                                 dtype=position_ids.dtype)[None]
        out_pos = _pos_skipping(self, out_pos, emb)
        return None, out_pos, out_mask, past_key_values, emb, out_labels


def _pos_skipping(lm, position_ids, emb):
    """`use_pos_skipping` (llava_arch.py:869-875, training only): position ids 0..L-1 with a random offset in front of
    and behind a random split point (long-context extension trick of the reference's trainer).  Same three draws from
    Python's `random`, in the same order."""
    if not (getattr(lm.config, "use_pos_skipping", False) and getattr(lm, "training", False)):
        return position_ids
    L = emb.size(1)
    position_ids = torch.arange(L, device=emb.device).unsqueeze(0)
    split_position = random.randint(0, L)
    left_add = random.randint(0, lm.config.pos_skipping_range)
    right_add = random.randint(left_add, lm.config.pos_skipping_range)
    position_ids[:, :split_position] += left_add
    position_ids[:, split_position:] += right_add
    return position_ids


def splice_into_text(lm, model, image_features: List[torch.Tensor], input_ids, position_ids, attention_mask,
                     past_key_values, labels):
    """Insert the visual token blocks at IMAGE_TOKEN_INDEX, truncate to tokenizer_model_max_length, pad, and build
    labels / mask / position ids (llava_arch.py:745-878).  Host-side orchestration of device copies."""
    _labels, _position_ids, _attention_mask = labels, position_ids, attention_mask
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    else:
        attention_mask = attention_mask.bool()
    if position_ids is None:
        position_ids = torch.arange(0, input_ids.shape[1], dtype=torch.long, device=input_ids.device)
    if labels is None:
        labels = torch.full_like(input_ids, IGNORE_INDEX)
    ids_list = [i[m] for i, m in zip(input_ids, attention_mask)]
    lab_list = [l[m] for l, m in zip(labels, attention_mask)]

    new_embeds, new_labels = [], []
    img_i = 0
    for ids, lab in zip(ids_list, lab_list):
        img_pos = torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist()
        if not img_pos:
            new_embeds.append(model.embed_tokens(ids))
            new_labels.append(lab)
            img_i += 1
            continue
        cuts = [-1] + img_pos + [ids.shape[0]]
        text_ids = [ids[cuts[i] + 1:cuts[i + 1]] for i in range(len(cuts) - 1)]
        text_lab = [lab[cuts[i] + 1:cuts[i + 1]] for i in range(len(cuts) - 1)]
        text_emb = torch.split(model.embed_tokens(torch.cat(text_ids)), [t.shape[0] for t in text_ids], dim=0)
        pe, pl = [], []
        for i in range(len(img_pos) + 1):
            pe.append(text_emb[i])
            pl.append(text_lab[i])
            if i < len(img_pos):
                f = image_features[min(img_i, len(image_features) - 1)]
                img_i += 1
                pe.append(f.to(text_emb[i].dtype))
                pl.append(torch.full((f.shape[0],), IGNORE_INDEX, device=lab.device, dtype=lab.dtype))
        new_embeds.append(torch.cat(pe))
        new_labels.append(torch.cat(pl))

    max_tok = getattr(lm.config, "tokenizer_model_max_length", None)
    new_embeds = [x[:max_tok] for x in new_embeds]
    new_labels = [x[:max_tok] for x in new_labels]
    max_len = max(x.shape[0] for x in new_embeds)
    B = len(new_embeds)
    left = getattr(lm.config, "tokenizer_padding_side", "right") == "left"
    emb = torch.zeros((B, max_len, new_embeds[0].shape[1]), dtype=new_embeds[0].dtype, device=new_embeds[0].device)
    lab_pad = torch.full((B, max_len), IGNORE_INDEX, dtype=new_labels[0].dtype, device=new_labels[0].device)
    mask = torch.zeros((B, max_len), dtype=attention_mask.dtype, device=attention_mask.device)
    pos = torch.zeros((B, max_len), dtype=position_ids.dtype, device=position_ids.device)
    for i, (e, l) in enumerate(zip(new_embeds, new_labels)):
        n = e.shape[0]
        if n == 0:
            continue
        sl = slice(max_len - n, max_len) if left else slice(0, n)
        emb[i, sl] = e
        lab_pad[i, sl] = l
        mask[i, sl] = True
        pos[i, sl] = torch.arange(0, n, dtype=pos.dtype, device=pos.device)
    out_labels = None if _labels is None else lab_pad
    out_mask = None if _attention_mask is None else mask.to(dtype=_attention_mask.dtype)
    out_pos = _pos_skipping(lm, None if _position_ids is None else pos, emb)
    return None, out_pos, out_mask, past_key_values, emb, out_labels
