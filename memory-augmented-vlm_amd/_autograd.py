"""Training path of the memory modules: torch.autograd.Function wrappers whose forward AND
backward are the HIP kernels of this library.  The reference trains the same modules with plain autograd
(`recurrent_model` / `larimar_model` are the trainable parts, train.py:1708-1724); autograd here only records the
graph and routes gradients - every product, softmax, normalisation and activation runs in csrc/.

The forward of each Function issues exactly the kernel sequence of the fused inference step (`mavlm_step`,
csrc/mavlm_api.hip), so training-mode activations are bit-identical to inference-mode ones (tested).
Gradients are produced in the parameter dtype (bf16 / fp16), as autograd does for the reference under
`--bf16 True`; reductions (dW contraction, bias / LayerNorm-affine sums) accumulate in fp32 before the one rounding.
"""
import math

import torch

from . import _capi as capi
from . import _ops as ops

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_WT = {}        # (data_ptr, shape, dtype, stream) -> (version, W^T): the transposed weights of ONE backward pass


def _wT(w):
    """W^T [K, N] for dx = dy W (the GEMM wants both operands contraction-contiguous).  A weight used by several steps of a
    video (the projector runs once per chunk) is transposed once per backward pass: the cache is emptied by every forward of
    these functions - weights change between a backward pass and the next forward, never inside a backward pass."""
    key = (w.data_ptr(), tuple(w.shape), w.dtype, torch.cuda.current_stream().cuda_stream)
    hit = _WT.get(key)
    if hit is not None and hit[0] == w._version:
        return hit[1]
    t = ops.transpose(w)
    _WT[key] = (w._version, t)
    return t


def _weight_grads(dy, x):
    """dW = dy^T x and db = column sums of dy, both through the transposed (contraction-contiguous) operands."""
    dyT = ops.transpose(dy)                     # [N, Mpad], zero pad
    xT = ops.transpose(x)                       # [K, Mpad]
    return ops.matmul_nt_splitk(dyT, xT), ops.rowsum(dyT, dy.shape[0])


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b)   (nn.Linear + ReLU / GELU; MemoryController.py:37-39,48-50,63-64; llava_arch.py:132-136)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        _WT.clear()
        b32 = bias.detach().float()
        x = _c(x.detach())
        w = weight.detach()
        if act == ACT_GELU:
            # the pre-activation is stored (its derivative needs it); the forward value is gelu(16-bit pre-activation),
            # i.e. one rounding more than the inference epilogue - the reference's bf16 graph rounds there as well
            pre = ops.linear(x, w, b32, capi.EPI_BIAS)
            y = ops.act(ops.ACT_GELU, pre)
            ctx.save_for_backward(x, w, pre)
        elif act == ACT_RELU:
            y = ops.linear(x, w, b32, capi.EPI_RELU)
            ctx.save_for_backward(x, w, y)
        else:
            y = ops.linear(x, w, b32, capi.EPI_BIAS)
            ctx.save_for_backward(x, w)
        ctx.act = act
        ctx.bias_dtype = bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        x, w = saved[0], saved[1]
        dy = _c(dy)
        if ctx.act == ACT_GELU:
            dy = ops.act(ops.ACT_GELU_BWD, saved[2], dy)
        elif ctx.act == ACT_RELU:
            dy = ops.act(ops.ACT_RELU_BWD, saved[2], dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.matmul_nt(dy, _wT(w))                    # dy [M,N] . (W^T [K,N])^T
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw, db = _weight_grads(dy, x)
            db = db.to(ctx.bias_dtype)
        return dx, dw, db, None


class DenseResidualNormFn(torch.autograd.Function):
    """y = LayerNorm(x W^T + b + res)   (`Residual`, MemoryController.py:20-29): fp32 GEMM output -> fused
    residual-add + LayerNorm kernel, as in the inference step."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, gamma, beta, eps):
        _WT.clear()
        x, res, w = _c(x.detach()), _c(res.detach()), weight.detach()
        g32 = gamma.detach().float()
        # the form the fused inference step takes for this shape (one kernel where the GEMM fills the chip, else GEMM + row
        # LayerNorm): bit-identical activations; z = x W^T + b in fp32 is what the backward recomputes the statistics from
        y, z = ops.linear_residual_layernorm(x, w, bias.detach().float(), res, g32, beta.detach().float(), eps, want_pre=True)
        ctx.save_for_backward(x, w, z, res, g32)
        ctx.eps = eps
        ctx.dtypes = (bias.dtype, gamma.dtype, beta.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, z, res, g32 = ctx.saved_tensors
        dz, dg, dbeta = ops.layernorm_bwd(_c(dy), z, res, g32, ctx.eps)
        dx = ops.matmul_nt(dz, _wT(w)) if ctx.needs_input_grad[0] else None
        dw, db = _weight_grads(dz, x)
        bd, gd, btd = ctx.dtypes
        return dx, dw, db.to(bd), (dz if ctx.needs_input_grad[3] else None), dg.to(gd), dbeta.to(btd), None


class AttentionFn(torch.autograd.Function):
    """ctx = softmax(q k^T * scale) v per head (MemoryController.py:51-54), heads of 128 columns (narrower heads are
    zero-padded by the caller).  Saves O and the log-sum-exp; the backward recomputes the probabilities per tile."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale, head_dim=128, plain=False):
        q, k, v = q.detach(), k.detach(), v.detach()
        o, lse = ops.attention(q, k, v, heads, want_lse=True, scale=scale, head_dim=head_dim, plain=plain)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.heads, ctx.scale, ctx.head_dim = heads, scale, head_dim
        ctx.mark_non_differentiable(lse)
        return o, lse

    @staticmethod
    def backward(ctx, do, _dlse):
        q, k, v, o, lse = ctx.saved_tensors
        n = ctx.needs_input_grad
        if ctx.head_dim == 128:
            dq, dk, dv = ops.attention_bwd(q, k, v, o, _c(do), lse, ctx.heads, n[0], n[1], n[2], scale=ctx.scale)
        else:       # wide heads (448): one head's scores at a time, every product a GEMM (ops.attention_bwd_wide)
            dq, dk, dv = ops.attention_bwd_wide(q, k, v, o, _c(do), lse, ctx.heads, ctx.head_dim, ctx.scale, n[0], n[1], n[2])
        return dq, dk, dv, None, None, None, None


def head_width(hd):
    """Columns one head occupies in the projected Q/K/V layout: narrow heads are zero-padded to 128, wide ones (448,
    the wide-head kernels) are not padded."""
    return 128 if hd <= 128 else hd


def pad_heads_out(t, heads, hd):
    """[H*hd, ...] -> [H*128, ...]: zero rows after each head (projection OUTPUT side); autograd-transparent."""
    if hd >= 128:
        return t
    tail = t.shape[1:]
    return torch.nn.functional.pad(t.reshape(heads, hd, -1), (0, 0, 0, 128 - hd)).reshape(heads * 128, *tail)


def pad_heads_in(t, heads, hd):
    """[D, H*hd] -> [D, H*128]: zero columns after each head (projection INPUT side)."""
    if hd >= 128:
        return t
    return torch.nn.functional.pad(t.reshape(t.shape[0], heads, hd), (0, 128 - hd)).reshape(t.shape[0], heads * 128)


def packed_kv_params(attns):
    """[k_0; v_0; k_1; v_1; ...] projection weights / biases of several Attention modules, heads zero-padded to 128
    rows each - the packed layout of the inference step (w_kv_seg / w_kv_evo, MemoryController._Engine.pack), built
    with autograd-transparent ops so the gradients flow back to the individual nn.Linear parameters."""
    ws, bs = [], []
    for a in attns:
        H, hd = a.num_attention_heads, a.attention_head_size
        for lin in (a.k_proj, a.v_proj):
            ws.append(pad_heads_out(lin.weight, H, hd))
            bs.append(pad_heads_out(lin.bias, H, hd))
    return torch.cat(ws, dim=0), torch.cat(bs, dim=0)


def project_kv(attns, x):
    """K/V of `x` for all `attns` in ONE GEMM (as mavlm_step does): returns [(k, v), ...] column views of the packed
    [rows, 2*len(attns)*H*128] output."""
    w, b = packed_kv_params(attns)
    kv = LinearFn.apply(x, w.to(x.dtype), b, ACT_NONE)
    Dp = attns[0].num_attention_heads * head_width(attns[0].attention_head_size)
    return [(kv[:, (2 * i) * Dp:(2 * i + 1) * Dp], kv[:, (2 * i + 1) * Dp:(2 * i + 2) * Dp]) for i in range(len(attns))]


def attention_block(attn, q_in, k, v, want_stats=False, patches_per_frame=196):
    """`Attention.forward` (MemoryController.py:47-56) on a 2-D [rows, D] query input and already projected K/V
    (padded-head layout; may be column views); returns (out, (q, k, lse) | None)."""
    H = attn.num_attention_heads
    hd = attn.attention_head_size
    if hd > 128 and hd != 448:
        raise capi.MavlmError(f"head_dim {hd}: no attention kernel (<= 128, or 448)")
    scale = ops.attn_scale(hd)
    dt = q_in.dtype      # parameters kept in fp32 (master weights) are cast per use; the cast is autograd-transparent
    q = LinearFn.apply(q_in, pad_heads_out(attn.q_proj.weight, H, hd).to(dt), pad_heads_out(attn.q_proj.bias, H, hd),
                       ACT_NONE)
    # (the fused step's last formation layer carries the frame scores on the SAME schedule as every other attention of
    # the shape - round 3 - so there is no special case here: `ops.attention` takes that schedule)
    ctxv, lse = AttentionFn.apply(q, k, v, H, scale, head_width(hd), False)
    d = attn.residual
    out = DenseResidualNormFn.apply(ctxv, pad_heads_in(d.dense.weight, H, hd).to(dt), d.dense.bias, q_in,
                                    d.layernorm.weight, d.layernorm.bias, d.layernorm.eps)
    return out, ((q, k, lse) if want_stats else None)


def mlp_block(layer, a):
    """TransformerLayer's MLP + Residual (MemoryController.py:63-67,71)."""
    up = layer.mlp[0]
    h = LinearFn.apply(a, up.weight.to(a.dtype), up.bias, ACT_RELU)
    d = layer.residual
    return DenseResidualNormFn.apply(h, d.dense.weight.to(a.dtype), d.dense.bias, a, d.layernorm.weight, d.layernorm.bias,
                                     d.layernorm.eps)


def fuser_mlp(fuser, x, type_row=None):
    """memory_fuser = Linear GELU Linear (llava_arch.py:132-136,546); `type_row` (token_type_embedding row 0) rides in
    the second bias so the add costs no extra rounding (as the inference epilogue, :548-553)."""
    shp = x.shape
    u = LinearFn.apply(x.reshape(-1, shp[-1]), fuser[0].weight.to(x.dtype), fuser[0].bias, ACT_GELU)
    b2 = fuser[2].bias.float()
    if type_row is not None:
        b2 = b2 + type_row.float()
    return LinearFn.apply(u, fuser[2].weight.to(x.dtype), b2, ACT_NONE).reshape(shp)
