"""Recurrent memory transformer on the MI355X HIP path.

Drop-in for llava/model/memory_module/MemoryController.py of the reference: same class names, constructor
arguments, parameter / state-dict names and the same call protocol

    recurrent_model.memory_cache = []                      # per video   (llava_arch.py:532)
    memory_cache, attn_stats = recurrent_model(segment)    # per chunk   (llava_arch.py:536-537)

but ``TransformerProjector.forward`` runs as one fused launch sequence of hand-written gfx950 kernels behind the
C ABI (include/mavlm.h: mavlm_step) - MFMA GEMMs with fused bias/ReLU/residual epilogues, flash-style
cross-attention that never materialises the [H,R,S] probabilities, a second column-sum pass for the frame scores,
wave-reduce LayerNorm.  The sub-modules keep their parameters (so checkpoints load unchanged) and each has a working
``forward`` built from the operator-level entry points; none of them has a CPU path.

Deliberate deviations from the reference (documented in DESIGN.md):
  * `Attention.forward` returns a `FusedAttentionStats` (column sums + log-sum-exp) instead of the
    materialised probability tensor (MemoryController.py:52,57) - the only consumer reads column sums (:135).
  * the dead per-chunk statistics of the evolution step (:99-109) are not computed, so the reference's implicit
    `num_memory_tokens % 8 == 0` / `heads == 8` requirement (reshape(8,-1,8), :109) does not apply.
  * tensors in the returned `memory_cache` are views into a ring buffer owned by the module: an entry is
    overwritten once it has been evicted from the FIFO (the reference drops the evicted tensor, :153-154).
  * with autograd recording and trainable parameters the same kernels run as autograd Functions (_autograd.py,
    DESIGN.md §9); the forward activations are bit-identical to the inference path.
"""
import math
from typing import List

import torch
from torch import nn

from ... import _capi as capi
from ... import _ops as ops


class Config:
    """Hyper-parameters; same attribute names and defaults as the reference `Config` (:7-18) plus the two
    constants the reference hard-codes in code (FIFO cap :153, chunk size llava_arch.py:528)."""
    mm_hidden_size = 896
    mm_hidden_act = "relu"
    mm_num_attention_heads = 8
    patch_size = 196
    mm_attention_probs_dropout_prob = 0.1   # defined, never applied (as in the reference)
    mm_layer_norm_eps = 1e-12
    mm_hidden_dropout_prob = 0.1            # defined, never applied
    mm_intermediate_size = 4 * mm_hidden_size
    num_memory_tokens = 8
    depth = 1
    mm_dtype = torch.float16                # creation dtype of the parameters only
    cache_cap = 10
    max_chunk_frames = 32


def _wants_grad(module, *tensors) -> bool:
    """Autograd is recording and something on this call can receive a gradient: take the training path (the same
    kernels wrapped as autograd Functions, _autograd.py) instead of the plain launches, which record no graph."""
    return torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors)
                                        or any(p.requires_grad for p in module.parameters()))


class FusedAttentionStats:
    """What the fused kernel keeps of the attention probabilities: per-(head,query) log2-sum-exp and, on demand,
    the per-key column sums  sum_h sum_q p[h,q,k]  (all the reference ever reads, MemoryController.py:135)."""

    def __init__(self, q, k, lse2, heads):
        self._q, self._k, self.lse2, self.heads = q, k, lse2, heads

    def column_sums_per_head(self) -> torch.Tensor:
        return ops.attention_colsum(self._q, self._k, self.lse2, self.heads)

    def column_sums(self) -> torch.Tensor:
        return self.column_sums_per_head().sum(dim=0)


class Residual(nn.Module):
    """LayerNorm(dense(h) + x)  (:20-29): GEMM with fused bias+residual epilogue (fp32 out) -> row LayerNorm."""

    def __init__(self, input_size, output_size, config):
        super().__init__()
        self.dense = nn.Linear(input_size, output_size, dtype=config.mm_dtype)
        self.layernorm = nn.LayerNorm(output_size, eps=config.mm_layer_norm_eps, dtype=config.mm_dtype)

    def forward(self, hidden_states: torch.Tensor, input_tensor: torch.Tensor):
        shp = input_tensor.shape
        h2 = hidden_states.reshape(-1, hidden_states.shape[-1])
        x2 = input_tensor.reshape(-1, shp[-1])
        if _wants_grad(self, hidden_states, input_tensor):
            from ... import _autograd as ag
            return ag.DenseResidualNormFn.apply(h2, self.dense.weight.to(h2.dtype), self.dense.bias, x2,
                                                self.layernorm.weight, self.layernorm.bias, self.layernorm.eps).reshape(shp)
        out, _ = ops.linear_residual_layernorm(h2, self.dense.weight, self.dense.bias.float(), x2.contiguous(),
                                               self.layernorm.weight.float(), self.layernorm.bias.float(), self.layernorm.eps)
        return out.reshape(shp)


class Attention(nn.Module):
    """Multi-head attention block (:31-57), head_dim 128 on the HIP path."""

    def __init__(self, config):
        super().__init__()
        self.hidden_size = config.mm_hidden_size
        self.num_attention_heads = config.mm_num_attention_heads
        self.attention_head_size = self.hidden_size // self.num_attention_heads
        self.k_proj = nn.Linear(self.hidden_size, self.hidden_size, dtype=config.mm_dtype)
        self.v_proj = nn.Linear(self.hidden_size, self.hidden_size, dtype=config.mm_dtype)
        self.q_proj = nn.Linear(self.hidden_size, self.hidden_size, dtype=config.mm_dtype)
        self.residual = Residual(self.hidden_size, self.hidden_size, config)

    def forward(self, hidden_states, kv_hidden_states=None, output_attentions=True):
        if hidden_states.dim() != 3 or hidden_states.shape[0] != 1:
            raise capi.MavlmError("Attention.forward: batch size 1 only (as the reference, MemoryController.py:121)")
        if self.attention_head_size != 128:
            raise capi.MavlmError("stand-alone Attention.forward needs head_dim 128; head_dim < 128 runs through the "
                                  "fused TransformerProjector path (zero-padded heads)")
        kv = hidden_states if kv_hidden_states is None else kv_hidden_states
        xq, xkv = hidden_states[0], kv[0]
        if _wants_grad(self, hidden_states, kv):
            from ... import _autograd as ag
            k_, v_ = ag.project_kv([self], xkv)[0]
            out, st = ag.attention_block(self, xq, k_, v_, want_stats=output_attentions)
            stats = FusedAttentionStats(st[0].detach(), st[1].detach(), st[2], self.num_attention_heads) if st else None
            return out[None], stats
        q = ops.linear(xq, self.q_proj.weight, self.q_proj.bias.float())
        k = ops.linear(xkv, self.k_proj.weight, self.k_proj.bias.float())
        v = ops.linear(xkv, self.v_proj.weight, self.v_proj.bias.float())
        ctx, lse2 = ops.attention(q, k, v, self.num_attention_heads, want_lse=output_attentions)
        out = self.residual(ctx[None], hidden_states)
        stats = FusedAttentionStats(q, k, lse2, self.num_attention_heads) if output_attentions else None
        return out, stats


class TransformerLayer(nn.Module):
    """Cross-attention + ReLU MLP with post-LN residuals (:59-72)."""

    def __init__(self, config):
        super().__init__()
        if config.mm_hidden_act != "relu":
            raise capi.MavlmError("only mm_hidden_act='relu' (the value the reference uses) is implemented")
        self.memory_segment_fusion_attention = Attention(config)
        self.mlp = nn.Sequential(
            nn.Linear(config.mm_hidden_size, config.mm_intermediate_size, dtype=config.mm_dtype),
            nn.ReLU(),
        )
        self.residual = Residual(config.mm_intermediate_size, config.mm_hidden_size, config)

    def forward(self, query_states, kv_states):
        a, stats = self.memory_segment_fusion_attention(query_states, kv_hidden_states=kv_states, output_attentions=True)
        if _wants_grad(self, query_states, kv_states):
            from ... import _autograd as ag
            return ag.mlp_block(self, a[0])[None], stats
        h = ops.linear(a[0], self.mlp[0].weight, self.mlp[0].bias.float(), capi.EPI_RELU)
        return self.residual(h[None], a), stats


class _Engine:
    """Owns the C handle, the packed parameter views and the device buffers of one TransformerProjector."""

    def __init__(self, proj: "TransformerProjector", device, dtype, max_chunk_frames, batch: int = 1, shard=None,
                 fused_ln_never: bool = False):
        cfg = proj.config
        self.device, self.dtype = device, dtype
        self.batch = int(batch)
        self.shard = tuple(shard) if shard else None      # (first memory token, tokens) this engine computes: row shard
        self.c = capi.Config(hidden=cfg.mm_hidden_size, heads=cfg.mm_num_attention_heads, patches=cfg.patch_size,
                             mem_tokens=cfg.num_memory_tokens, depth=cfg.depth, inter=cfg.mm_intermediate_size,
                             cache_cap=getattr(cfg, "cache_cap", 10), max_chunk_frames=max_chunk_frames,
                             dtype=ops.dtype_code(dtype), eps=cfg.mm_layer_norm_eps, batch=self.batch,
                             q_token0=self.shard[0] if self.shard else 0, q_tokens=self.shard[1] if self.shard else 0,
                             fused_ln=capi.LN_NEVER if fused_ln_never else capi.LN_AUTO)
        lib = capi.lib()
        h = capi.vp()
        capi.check(lib.mavlm_create(self.c, h), "mavlm_create")
        self.ctx = h
        R, D = self.c.mem_tokens * self.c.patches, self.c.hidden
        self.head_dim = D // self.c.heads
        # attention buffer width: heads zero-padded to 128 columns, or native for the wide-head kernel (448, OV-7B)
        self.Dp = D if self.head_dim > 128 else self.c.heads * 128
        # row batch (batch > 1): a FIFO slot holds the memory of every video ([cap, B, M, P, D] - one contiguous GEMM
        # operand per slot), the evolution K/V of ONE video are contiguous over its slots ([B, cap, R, 2Dp]); include/mavlm.h
        if self.batch > 1:
            self.mem_ring = torch.empty((self.c.cache_cap, self.batch, self.c.mem_tokens, self.c.patches, D), device=device, dtype=dtype)
            self.evo_kv = torch.empty((self.batch, self.c.cache_cap, R, 2 * self.Dp), device=device, dtype=dtype)
        else:
            self.mem_ring = torch.empty((self.c.cache_cap, self.c.mem_tokens, self.c.patches, D), device=device, dtype=dtype)
            self.evo_kv = torch.empty((self.c.cache_cap, R, 2 * self.Dp), device=device, dtype=dtype)
        nbytes = lib.mavlm_workspace_bytes(self.c)
        # (the exchange scratch of the fused LayerNorm epilogue inside it is zero-filled by the library at the first step)
        self.workspace = torch.empty(nbytes + 256, device=device, dtype=torch.uint8)
        base = (self.workspace.data_ptr() + 255) & ~255
        self.workspace_base_offset = base - self.workspace.data_ptr()
        b = capi.Buffers(mem_ring=self.mem_ring.data_ptr(), evo_kv_ring=self.evo_kv.data_ptr(), workspace=base,
                         workspace_bytes=nbytes)
        capi.check(lib.mavlm_bind_buffers(self.ctx, b), "mavlm_bind_buffers")
        self.keep = {}
        self.version = None
        self._probe = None            # pinned host copy of the exchange control words (mavlm_ln_status_async)
        self._probe_evt = None
        self.ln_timeouts_seen = 0

    # -- health of the fused Residual kernel's exchange (include/mavlm.h: MAVLM_LN_MAX_STREAMS) ----------------------------
    def check_ln_probe(self, wait: bool = False):
        """Looks at the last posted probe if its copy has landed (never blocks unless `wait`); raises MavlmError when a launch
        of the fused dense + residual + LayerNorm kernel gave up waiting for a partner workgroup since the probe before."""
        evt = self._probe_evt
        if evt is None or torch.cuda.is_current_stream_capturing():     # (an event query is illegal inside a graph capture)
            return
        if wait:
            evt.synchronize()
        elif not evt.query():
            return
        self._probe_evt = None
        t = int(self._probe[2])
        if t:
            self.ln_timeouts_seen += t
            raise capi.MavlmError("fused dense + residual + LayerNorm kernel: a workgroup timed out waiting for its row block's "
                                  "partners - the memory computed since the previous video is wrong (more than "
                                  f"{capi.LN_MAX_STREAMS} streams on the fused form, or a stalled device); the flag has been "
                                  "cleared, re-run the video")

    def post_ln_probe(self):
        """One asynchronous 16-byte read of the control words per video (called at `memory_cache = []`), checked at the next
        one: the product path notices a timed-out exchange without ever synchronising."""
        self.check_ln_probe()
        if self._probe_evt is not None or torch.cuda.is_current_stream_capturing():
            return                                        # previous copy still in flight / never inside a graph capture
        if self._probe is None:
            self._probe = torch.zeros(4, dtype=torch.int32).pin_memory()
        rc = capi.lib().mavlm_ln_status_async(self.ctx, self._probe.data_ptr(), 1, ops.stream_ptr())
        if rc == 0:
            self._probe_evt = torch.cuda.Event()
            self._probe_evt.record()
        elif rc < 0:
            capi.check(rc, "mavlm_ln_status_async")

    def __del__(self):
        try:
            if self.ctx:
                capi.lib().mavlm_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # -- parameter packing -------------------------------------------------------------------------------
    def pack(self, proj, fuser=None, type_emb=None):
        dev, dt = self.device, self.dtype
        keep = {}

        def w16(name, t):
            keep[name] = t.detach().to(device=dev, dtype=dt).contiguous()
            return keep[name].data_ptr()

        def f32(name, t):
            keep[name] = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            return keep[name].data_ptr()

        H, hd = self.c.heads, self.head_dim

        def pad_out(t):      # [H*hd, ...] -> [H*128, ...]: zero rows after each head (projection OUTPUT side)
            if hd >= 128:
                return t
            t = t.detach()
            shp = t.shape[1:]
            z = torch.zeros((H, 128) + tuple(shp), device=t.device, dtype=t.dtype)
            z[:, :hd] = t.reshape((H, hd) + tuple(shp))
            return z.reshape((H * 128,) + tuple(shp))

        def pad_in(t):       # [D, H*hd] -> [D, H*128]: zero columns after each head (projection INPUT side)
            if hd >= 128:
                return t
            t = t.detach()
            z = torch.zeros((t.shape[0], H, 128), device=t.device, dtype=t.dtype)
            z[:, :, :hd] = t.reshape(t.shape[0], H, hd)
            return z.reshape(t.shape[0], H * 128)

        def attn(prefix, a: Attention):
            return capi.AttnWeights(wq=w16(prefix + "wq", pad_out(a.q_proj.weight)), bq=f32(prefix + "bq", pad_out(a.q_proj.bias)),
                                    wo=w16(prefix + "wo", pad_in(a.residual.dense.weight)), bo=f32(prefix + "bo", a.residual.dense.bias),
                                    ln_g=f32(prefix + "g", a.residual.layernorm.weight),
                                    ln_b=f32(prefix + "b", a.residual.layernorm.bias))

        W = capi.Weights()
        R, D = self.c.mem_tokens * self.c.patches, self.c.hidden
        # initial_memory + memory_pos_embed in the parameter dtype, then cast (MemoryController.py:123-124)
        mem0 = (proj.initial_memory + proj.memory_pos_embed).reshape(R, D)
        if self.shard:                                          # row shard: the owned rows only
            mem0 = mem0[self.shard[0] * self.c.patches:(self.shard[0] + self.shard[1]) * self.c.patches]
        W.mem0 = w16("mem0", mem0.repeat(self.batch, 1) if self.batch > 1 else mem0)     # [B*R, D]: every video starts from it
        ats = [l.memory_segment_fusion_attention for l in proj.layers]
        W.w_kv_seg = w16("wkv", torch.cat([pad_out(t) for a in ats for t in (a.k_proj.weight, a.v_proj.weight)], dim=0))
        W.b_kv_seg = f32("bkv", torch.cat([pad_out(t) for a in ats for t in (a.k_proj.bias, a.v_proj.bias)], dim=0))
        for l, layer in enumerate(proj.layers):
            W.layer_attn[l] = attn(f"l{l}.", ats[l])
            W.w_up[l] = w16(f"l{l}.up", layer.mlp[0].weight)
            W.b_up[l] = f32(f"l{l}.bup", layer.mlp[0].bias)
            W.w_down[l] = w16(f"l{l}.down", layer.residual.dense.weight)
            W.b_down[l] = f32(f"l{l}.bdown", layer.residual.dense.bias)
            W.ln2_g[l] = f32(f"l{l}.g2", layer.residual.layernorm.weight)
            W.ln2_b[l] = f32(f"l{l}.b2", layer.residual.layernorm.bias)
        e = proj.memory_update_attention
        W.evo = attn("evo.", e)
        W.w_kv_evo = w16("evo.wkv", torch.cat([pad_out(e.k_proj.weight), pad_out(e.v_proj.weight)], dim=0))
        W.b_kv_evo = f32("evo.bkv", torch.cat([pad_out(e.k_proj.bias), pad_out(e.v_proj.bias)], dim=0))
        if fuser is not None and type_emb is not None:
            te = type_emb.weight.detach().to(device=dev, dtype=dt)
            W.w_f1 = w16("f1", fuser[0].weight)
            W.b_f1 = f32("bf1", fuser[0].bias)
            W.w_f2 = w16("f2", fuser[2].weight)
            # bias + token_type_embedding[0] folded into the second epilogue (llava_arch.py:548-553)
            W.b_f2_type0 = f32("bf2", fuser[2].bias.detach().to(dev).float() + te[0].float())
            W.type1 = w16("type1", te[1])
        capi.check(capi.lib().mavlm_bind_weights(self.ctx, W), "mavlm_bind_weights")
        self.keep = keep          # keeps the packed tensors (and the Weights struct's targets) alive
        self._W = W

    @property
    def steps(self) -> int:
        return capi.lib().mavlm_steps(self.ctx)

    def ln_exchange_status(self):
        """(launches, timeouts) of the fused dense + residual + LayerNorm epilogue in this engine's workspace, or None when the
        config never takes the fused form.  timeouts != 0: a workgroup gave up waiting for a partner (bounded spin) and the
        result of that launch is wrong - never seen; the tests and bench.py assert 0.  Synchronises (a D2H read)."""
        off = int(capi.lib().mavlm_ln_ctl_offset(self.ctx))
        if off < 0:
            return None
        w = self.workspace[self.workspace_base_offset + off:self.workspace_base_offset + off + 16].view(torch.int32).cpu()
        if self._probe_evt is not None:                   # a posted probe clears the device flag: fold it in
            self._probe_evt.synchronize()
            self._probe_evt = None
            self.ln_timeouts_seen += int(self._probe[2])
        return int(w[1]), int(w[2]) + self.ln_timeouts_seen

    def workspace_views(self):
        """Typed views of the workspace regions (contents = intermediates of the most recent sub-layer).  For the
        stage-wise parity tests; not used by the product path."""
        import ctypes
        offs = (ctypes.c_size_t * 10)()
        capi.check(capi.lib().mavlm_workspace_layout(self.c, offs, 10), "mavlm_workspace_layout")
        c = self.c
        R, S, D, I, L, H = c.mem_tokens * c.patches, c.max_chunk_frames * c.patches, c.hidden, c.inter, c.depth, c.heads
        names = ("kv_seg", "q", "ctx", "a", "h", "pre", "mA", "mB", "lse2", "part")
        Dp = self.Dp
        shapes = ((S, 2 * L * Dp), (R, Dp), (R, Dp), (R, D), (R, I), (R, D), (R, D), (R, D), (H, R), (H, S))
        dts = (self.dtype,) * 5 + (torch.float32, self.dtype, self.dtype, torch.float32, torch.float32)
        out = {}
        for n_, o, shp, dt in zip(names, offs, shapes, dts):
            nbytes = shp[0] * shp[1] * (4 if dt == torch.float32 else 2)
            b0 = self.workspace_base_offset + o
            out[n_] = self.workspace[b0:b0 + nbytes].view(dt).view(shp)
        return out

    def colsum_part(self, F):
        """[H, F*P] fp32 per-head column sums the last step with F frames left in the workspace: the column-sum pass writes
        one [H, S] plane per piece of its schedule (mavlm_attention_colsum_plan), added here in plane order."""
        import ctypes
        offs = (ctypes.c_size_t * 10)()
        capi.check(capi.lib().mavlm_workspace_layout(self.c, offs, 10), "mavlm_workspace_layout")
        c = self.c
        R, S, H = c.mem_tokens * c.patches, F * c.patches, c.heads
        planes = 1
        if self.Dp // H == 128:
            info = (ctypes.c_int32 * 2)()
            capi.check(capi.lib().mavlm_attention_colsum_plan(R, S, H, info), "mavlm_attention_colsum_plan")
            planes = info[1]
        b0 = self.workspace_base_offset + offs[9]
        flat = self.workspace[b0:b0 + planes * H * S * 4].view(torch.float32).view(planes, H, S)
        acc = flat[0].clone()
        for p_ in range(1, planes):
            acc += flat[p_]
        return acc


class TransformerProjector(nn.Module):
    """The recurrent memory transformer (:74-158)."""

    def __init__(self, config=None):
        super().__init__()
        self.config = config or Config()
        if self.config.depth > capi.MAX_DEPTH:
            raise capi.MavlmError(f"depth > {capi.MAX_DEPTH} not supported")
        self.layers = nn.ModuleList([TransformerLayer(self.config) for _ in range(self.config.depth)])
        self.num_memory_tokens = self.config.num_memory_tokens
        self.hidden_size = self.config.mm_hidden_size
        self.patch_size = self.config.patch_size
        self.initial_memory = nn.Parameter(torch.empty(self.num_memory_tokens, self.patch_size, self.hidden_size))
        self.memory_pos_embed = nn.Parameter(torch.randn(self.num_memory_tokens, 1, self.hidden_size))
        nn.init.xavier_uniform_(self.initial_memory)
        self.memory_update_attention = Attention(self.config)
        self.frame_attn_scores: List[torch.Tensor] = []
        self.compute_frame_scores = True      # API parity default; False skips the column-sum pass
        self._memory_cache: List[torch.Tensor] = []
        self._cache_mode = "engine"           # "engine": ring views (inference); "autograd": graph tensors (training)
        self._evo_kv = []                     # training path: (K, V) projections of the cached memories
        self._train_steps = 0                 # training path: memories produced since the last reset
        self._engine = None
        self._fuser_refs = None               # (memory_fuser, token_type_embedding) bound by the glue
        # Staleness of the engine's packed weight COPIES (fp32 biases / LayerNorm affines, mem0, concatenated and
        # head-padded matrices).  In-place updates through autograd-visible ops bump `p._version` and are seen by
        # _param_version(); updates through `.data` (DeepSpeed ZeRO-1/2 copy their bf16 partitions back with
        # `p.data.copy_()`) change neither the version nor the pointer.  Such updates only happen in training jobs, so:
        # every video that starts while the module is in training mode, or after it has been in training / autograd mode
        # since the last pack, re-packs.  One dict shared with the replicas of spawn_replica().
        self._train_state = {"epoch": 0, "training": self.training}
        self._packed_epoch = -1

    # -- the reference's reset protocol: `module.memory_cache = []` ----------------------------------------
    @property
    def memory_cache(self) -> List[torch.Tensor]:
        return self._memory_cache

    @memory_cache.setter
    def memory_cache(self, value):
        value = list(value)
        if value:
            raise capi.MavlmError("memory_cache can only be reset to [] from outside (llava_arch.py:532); "
                                  "its entries are views into the module's ring buffer")
        self._memory_cache = value
        self._evo_kv = []
        self._cache_mode = "engine"
        if self._engine is not None:
            capi.check(capi.lib().mavlm_reset(self._engine.ctx), "mavlm_reset")
            if self._weights_maybe_stale():
                self._engine.version = None          # re-pack on the first step of this video
            self._engine.post_ln_probe()             # raises if the previous videos' fused LayerNorm exchange timed out

    def train(self, mode: bool = True):
        st = self.__dict__.get("_train_state")
        if st is not None:
            st["training"] = bool(mode)
            if mode:
                st["epoch"] += 1
        return super().train(mode)

    def _weights_maybe_stale(self, packed_epoch=None) -> bool:
        """`packed_epoch`: the training epoch at which the asking engine packed its copies (default: this module's own engine;
        a BatchedProjector / RowShardedMemory keeps its own - round 4: they used to compare with the BASE module's, which
        never packs when only pools run, and re-packed the weights at every video)."""
        st = self._train_state
        fuser_training = any(m.training for m in (self._fuser_refs or ()))
        return st["training"] or fuser_training or st["epoch"] != (self._packed_epoch if packed_epoch is None else packed_epoch)

    # -- engine management -------------------------------------------------------------------------------
    def bind_fuser(self, memory_fuser, token_type_embedding):
        """Give the engine the Memory-Fuser MLP and the token-type embedding (llava_arch.py:132-136,150) so that
        mavlm_fuse_emit can run fuser + type add + concat in one sequence."""
        self._fuser_refs = (memory_fuser, token_type_embedding)
        if self._engine is not None:
            self._engine.version = None

    def _param_version(self):
        ps = list(self.parameters())
        if self._fuser_refs is not None:
            ps += list(self._fuser_refs[0].parameters()) + list(self._fuser_refs[1].parameters())
        return tuple((p._version, p.data_ptr()) for p in ps)

    def refresh_weights(self):
        if self._engine is not None:
            self._engine.version = None

    def engine(self, device, dtype, frames=None) -> _Engine:
        need = max(int(getattr(self.config, "max_chunk_frames", 32)), int(frames or 0))
        e = self._engine
        if e is None or e.device != device or e.dtype != dtype or e.c.max_chunk_frames < need:
            if e is not None and e.steps:
                raise capi.MavlmError("device / dtype / chunk size changed in the middle of a video")
            e = self._engine = _Engine(self, device, dtype, need, fused_ln_never=self._fused_ln_never)
            self._memory_cache = []
        v = self._param_version()
        if e.version != v:
            fuser, temb = self._fuser_refs if self._fuser_refs is not None else (None, None)
            e.pack(self, fuser, temb)
            e.version = v
            self._packed_epoch = self._train_state["epoch"]
        return e

    def _apply(self, fn, *a, **k):   # .to() / .cuda() / .half(): parameters may move, packed pointers would dangle
        before = [(p.data_ptr(), p.dtype, p.device) for p in self.parameters()]
        out = super()._apply(fn, *a, **k)
        after = [(p.data_ptr(), p.dtype, p.device) for p in self.parameters()]
        # the reference calls `.to(self.device)` on this module on EVERY forward (llava_arch.py:530): a no-op move must
        # not throw the engine (packed weights, rings, workspace) away
        if before != after and getattr(self, "_engine", None) is not None:
            self._engine = None           # also ends a video in progress: its FIFO lived in the old engine
            self._memory_cache = []
            self._evo_kv = []
        return out

    _fused_ln_never = False       # replicas of a pool with more than LN_MAX_STREAMS streams: GEMM + row LayerNorm kernels only

    def spawn_replica(self, fused_ln_never: bool = False) -> "TransformerProjector":
        """A second recurrent state over the SAME parameters: shares every Parameter / sub-module object with this
        module (no weight copy) but owns its own engine, FIFO ring and workspace.  Used to keep several videos in
        flight on different HIP streams (MemoryPathPool in llava_arch.py): one video's partial-wave kernel tails
        are then filled by the other's kernels."""
        import copy
        r = copy.copy(self)                    # shallow: _parameters / _modules dicts are shared
        r._engine = None
        r._memory_cache = []
        r._evo_kv = []
        r.frame_attn_scores = []
        r._fused_ln_never = bool(fused_ln_never)
        return r

    # -- the next chunk's K/V projection beside this chunk's step (round 4) ----------------------------------
    _ahead_stream = None

    def ahead_ok(self, force: bool = False) -> bool:
        """Is it worth projecting the NEXT chunk's K/V on a side stream while this chunk's step runs?  Only where the step's own
        kernels leave most of the chip idle - few memory rows (fewer than 128 tiles of 256 x 256 in a D-wide GEMM over them) - AND
        the projection is long enough to matter: measured (`tools/diag_ahead_ab.py`, same process) +1.3 % at the OneVision-7B width
        with 8 memory tokens (20.31 against 20.57 ms per 256-frame video), +-0.4 % at D = 1024 (off there; two streams of single
        videos: -1.8 %).  Inference path only, not inside a graph capture.  `llava_arch.PROJECT_AHEAD = True` forces it on."""
        if self._cache_mode == "autograd" or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        rows = self.num_memory_tokens * self.patch_size
        return -(-rows // 256) * -(-self.hidden_size // 256) < 128 and (force or self.hidden_size >= 2048)

    def project_ahead(self, next_segment: torch.Tensor):
        """`mavlm_project_chunk_ahead`: call BEFORE `self(segment_t)` with segment t + 1 (the tensor the next call will be given).
        The library orders the two streams with events; results do not change."""
        F = next_segment.shape[0]
        eng = self.engine(next_segment.device, next_segment.dtype, F)
        x = next_segment.contiguous()
        cur = torch.cuda.current_stream()
        if self._ahead_stream is None or self._ahead_stream.device != x.device:
            self._ahead_stream = torch.cuda.Stream(device=x.device)
        side = self._ahead_stream
        side.wait_stream(cur)                      # (the producer of the frames; also everything of the steps before this one)
        capi.check(capi.lib().mavlm_project_chunk_ahead(eng.ctx, x.data_ptr(), F, side.cuda_stream), "mavlm_project_chunk_ahead")
        x.record_stream(side)

    # -- forward -----------------------------------------------------------------------------------------
    def forward(self, image_features: torch.Tensor):
        if image_features.dim() != 3:
            raise capi.MavlmError("TransformerProjector expects [F, P, D]")
        if not image_features.is_cuda:
            raise capi.MavlmError("TransformerProjector: input is not on a GPU; the HIP path has no CPU fallback")
        F, P, D = image_features.shape
        if P != self.patch_size or D != self.hidden_size:
            raise capi.MavlmError(f"expected [F,{self.patch_size},{self.hidden_size}], got {tuple(image_features.shape)}")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_autograd(image_features)
        if self._cache_mode == "autograd" and self._memory_cache:
            raise capi.MavlmError("memory_cache holds autograd tensors of a training-mode video: reset it "
                                  "(`memory_cache = []`) before running the inference path")
        self._cache_mode = "engine"
        eng = self.engine(image_features.device, image_features.dtype, F)
        x = image_features.contiguous()
        scores = torch.empty(F, device=x.device, dtype=x.dtype) if self.compute_frame_scores else None
        lib = capi.lib()
        capi.check(lib.mavlm_step(eng.ctx, x.data_ptr(), F, scores.data_ptr() if scores is not None else 0, 0,
                                  ops.stream_ptr()), "mavlm_step")
        self._memory_cache.append(eng.mem_ring[lib.mavlm_newest_slot(eng.ctx)])      # :152
        cap = eng.c.cache_cap
        if len(self._memory_cache) > cap:                                            # :153-154
            self._memory_cache = self._memory_cache[-cap:]
        if scores is not None:
            self.frame_attn_scores.append(scores)                                    # :156-157
        return self._memory_cache, self.frame_attn_scores

    def _forward_autograd(self, image_features: torch.Tensor):
        """Training path (the reference runs the same forward under autograd, BPTT through the
        un-detached memory_cache, :125-127,152): the step is composed of autograd Functions whose forward and backward
        are the HIP kernels (_autograd.py), in the kernel order of the fused inference step - the activations are
        bit-identical to mavlm_step's.  The FIFO holds graph-carrying tensors instead of ring views."""
        from ... import _autograd as ag
        if self._cache_mode == "engine" and self._memory_cache:
            raise capi.MavlmError("memory_cache holds ring views of an inference-mode video: reset it "
                                  "(`memory_cache = []`) before running the training path")
        self._cache_mode = "autograd"
        self._train_state["epoch"] += 1          # an optimizer step may follow: packed copies are suspect afterwards
        F, P, D = image_features.shape
        R = self.num_memory_tokens * P
        dt = image_features.dtype
        frames = image_features.detach().reshape(F * P, D)         # frame features carry no gradient (llava_arch.py:302)
        cap = int(getattr(self.config, "cache_cap", 10))
        if self._memory_cache:                                      # :125-127, 89-97
            # K/V of a cached memory are row-independent: each memory is projected ONCE, when it is the newest (one
            # packed GEMM, as mavlm_step does into its evo_kv ring), and the projections are concatenated
            evo = self.memory_update_attention
            last = self._memory_cache[-1].reshape(R, D)
            while len(self._evo_kv) < len(self._memory_cache):
                mem = self._memory_cache[len(self._evo_kv)].reshape(R, D)
                self._evo_kv.append(ag.project_kv([evo], mem)[0])
            # key order = slot order of the inference ring (memory g lives in slot g % cap), so that the attention sums
            # in the same order after the FIFO has wrapped as well
            first = self._train_steps - len(self._evo_kv)          # global index of the oldest cached memory
            order = sorted(range(len(self._evo_kv)), key=lambda i: (first + i) % cap)
            k = torch.cat([self._evo_kv[i][0] for i in order], dim=0)
            v = torch.cat([self._evo_kv[i][1] for i in order], dim=0)
            m, _ = ag.attention_block(evo, last, k, v)
        else:
            self._evo_kv = []
            self._train_steps = 0
            m = (self.initial_memory + self.memory_pos_embed).to(dt).reshape(R, D)      # :123-124
        stats = None
        atts = [layer.memory_segment_fusion_attention for layer in self.layers]
        kvs = ag.project_kv(atts, frames)                           # chunk K/V of all layers: one GEMM
        for li, layer in enumerate(self.layers):                   # :132-133
            last_layer = li == len(self.layers) - 1
            a, stats = ag.attention_block(atts[li], m, kvs[li][0], kvs[li][1],
                                          want_stats=last_layer and self.compute_frame_scores, patches_per_frame=P)
            m = ag.mlp_block(layer, a)
        self._memory_cache.append(m.reshape(self.num_memory_tokens, P, D))             # :152
        self._train_steps += 1
        if len(self._memory_cache) > cap:
            drop = len(self._memory_cache) - cap
            self._memory_cache = self._memory_cache[drop:]
            self._evo_kv = self._evo_kv[drop:]
        if stats is not None:                                      # :135-139,156 (detached statistics)
            with torch.no_grad():
                q, k, lse = stats
                att = self.layers[-1].memory_segment_fusion_attention
                hdw = 128 if att.attention_head_size <= 128 else att.attention_head_size
                part = ops.attention_colsum(q, k, lse, att.num_attention_heads, head_dim=hdw,
                                            scale=ops.attn_scale(att.attention_head_size))
                self.frame_attn_scores.append(part.sum(dim=0).view(F, P).mean(dim=1).to(dt))
        return self._memory_cache, self.frame_attn_scores


class BatchedProjector:
    """B independent videos stepped TOGETHER over the parameters of one `TransformerProjector` (row batch).

    The reference runs one video per forward (llava_arch.py:436) and every Linear / LayerNorm of the path sees the
    R = M*196 memory rows of that one video.  Those operators are row-independent and the weights are shared, so the rows
    of B videos stack into ONE [B*R, D] operand per launch (`mavlm_config.batch`): at the reference's M = 8 that turns
    1568-row GEMMs (7 of 256 CUs' worth of tiles) into chip-filling ones; the attention serves B*heads (video, head)
    pairs, each video over its own keys.  Videos of a batch step with the same chunk sizes (same length).

        bp = BatchedProjector(rm, 8)
        bp.reset()
        for chunk in chunks:  bp.step([x_b[lo:hi] for x_b in videos])      # x_b: [T, P, D] PE-added frames of video b
        bp.memory_cache(b) -> list of [M, P, D] ring views of video b;  bp.frame_scores -> list of [B, F] per chunk
    Inference only (no autograd path).  Same kernels and rounding points as the single-video engine; the attention's fp32
    summation order follows the schedule of the stacked grid."""

    def __init__(self, proj: TransformerProjector, batch: int, fused_ln_never: bool = False):
        self.fused_ln_never = bool(fused_ln_never)
        if batch < 2:
            raise capi.MavlmError("BatchedProjector: batch >= 2 (a single video runs through TransformerProjector itself)")
        self.proj, self.batch = proj, int(batch)
        self._engine = None
        self._packed_epoch = -1
        self.compute_frame_scores = True
        self.frame_scores: List[torch.Tensor] = []
        self._n = 0

    def engine(self, device, dtype, frames=None) -> _Engine:
        proj = self.proj
        need = max(int(getattr(proj.config, "max_chunk_frames", 32)), int(frames or 0))
        e = self._engine
        if e is None or e.device != device or e.dtype != dtype or e.c.max_chunk_frames < need:
            if e is not None and e.steps:
                raise capi.MavlmError("device / dtype / chunk size changed in the middle of a video batch")
            e = self._engine = _Engine(proj, device, dtype, need, batch=self.batch, fused_ln_never=self.fused_ln_never)
        v = proj._param_version()
        if e.version != v:
            fuser, temb = proj._fuser_refs if proj._fuser_refs is not None else (None, None)
            e.pack(proj, fuser, temb)
            e.version = v
            self._packed_epoch = proj._train_state["epoch"]
        return e

    def reset(self):
        """`memory_cache = []` for every video of the batch (llava_arch.py:532)."""
        self.frame_scores = []
        self._n = 0
        if self._engine is not None:
            capi.check(capi.lib().mavlm_reset(self._engine.ctx), "mavlm_reset")
            if self.proj._weights_maybe_stale(self._packed_epoch):
                self._engine.version = None
            self._engine.post_ln_probe()

    @torch.no_grad()
    def step(self, segs):
        """one chunk of every video: segs = B tensors [F, P, D] (contiguous, same F)"""
        if len(segs) != self.batch:
            raise capi.MavlmError(f"BatchedProjector.step: {self.batch} chunks expected")
        F, P, D = segs[0].shape
        for t in segs:
            if not t.is_cuda or tuple(t.shape) != (F, P, D) or not t.is_contiguous() or t.dtype != segs[0].dtype:
                raise capi.MavlmError("BatchedProjector.step: contiguous GPU chunks of one shape / dtype expected")
        if P != self.proj.patch_size or D != self.proj.hidden_size:
            raise capi.MavlmError(f"expected [F,{self.proj.patch_size},{self.proj.hidden_size}] chunks")
        eng = self.engine(segs[0].device, segs[0].dtype, F)
        ptrs = (capi.vp * self.batch)(*[t.data_ptr() for t in segs])
        scores = torch.empty((self.batch, F), device=segs[0].device, dtype=segs[0].dtype) if self.compute_frame_scores else None
        capi.check(capi.lib().mavlm_step_batch(eng.ctx, ptrs, F, scores.data_ptr() if scores is not None else 0, 0,
                                               ops.stream_ptr()), "mavlm_step_batch")
        self._n += 1
        if scores is not None:
            self.frame_scores.append(scores)
        return scores

    def memory_cache(self, b: int) -> List[torch.Tensor]:
        """video b's FIFO, oldest first (ring views, as TransformerProjector.memory_cache)"""
        eng = self._engine
        cap = eng.c.cache_cap
        n = min(self._n, cap)
        return [eng.mem_ring[(self._n - n + i) % cap, b] for i in range(n)]
