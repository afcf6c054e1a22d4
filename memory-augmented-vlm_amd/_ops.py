"""Operator-level wrappers over the C ABI (device pointers + current HIP stream).  PyTorch is used only to
own device memory and the stream.  Every function raises if the tensors are not on a GPU: there is no CPU path."""
import math

import torch

from . import _capi as capi


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return capi.BF16
    if dt == torch.float16:
        return capi.F16
    raise capi.MavlmError(f"dtype {dt} is not supported by the gfx950 kernels (bf16 / fp16 only); "
                          "the reference runs this path in the model dtype (bf16 in training, fp16/bf16 in eval)")


def attn_scale(head_dim: int) -> float:
    """1/sqrt(head_dim) exactly as the C side computes it (`1.0f / sqrtf((float)hd)`, csrc/mavlm_api.hip): float32
    square root, float32 division - so the training path and the fused inference step hand the kernels the same bits."""
    import numpy as np
    return float(np.float32(1.0) / np.sqrt(np.float32(head_dim)))


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise capi.MavlmError("tensor is not on a GPU: the memory path has no CPU fallback")


def _rows(x):
    if x.dim() != 2 or x.stride(1) != 1:
        raise capi.MavlmError("expected a 2-D tensor with unit inner stride")
    return x.shape[0], x.shape[1], x.stride(0)


def linear(x, weight, bias_f32, epilogue=capi.EPI_BIAS, residual=None, out=None):
    """out = epi(x @ weight.T + bias).  x [M,K], weight [N,K] (nn.Linear layout) 16-bit; bias fp32 [N].
    epilogue EPI_RES_F32 adds ``residual`` [M,N] and returns fp32."""
    _need_gpu(x, weight, bias_f32, residual, out)
    M, K, lda = _rows(x)
    N, K2, ldw = _rows(weight)
    if K2 != K or weight.dtype != x.dtype or bias_f32.dtype != torch.float32 or bias_f32.numel() != N:
        raise capi.MavlmError("linear: operand mismatch")
    if out is None:
        out = torch.empty((M, N), device=x.device,
                          dtype=torch.float32 if epilogue in (capi.EPI_RES_F32, capi.EPI_F32) else x.dtype)
    _, _, ldc = _rows(out)
    rp, ldr = (0, 0)
    if epilogue == capi.EPI_RES_F32:
        if residual is None or residual.dtype != x.dtype:
            raise capi.MavlmError("linear: residual required")
        _, _, ldr = _rows(residual)
        rp = residual.data_ptr()
    lib = capi.lib()
    nws = lib.mavlm_linear_ws_floats(M, N, K, epilogue, ldc)   # > 0: few tiles + long contraction -> split-K (as mavlm_step)
    ws = torch.empty((nws,), device=x.device, dtype=torch.float32) if nws else None
    capi.check(lib.mavlm_linear_ws(x.data_ptr(), lda, weight.data_ptr(), ldw, bias_f32.data_ptr(), rp, ldr,
                                   out.data_ptr(), ldc, M, N, K, epilogue, ws.data_ptr() if nws else 0, nws,
                                   dtype_code(x.dtype), stream_ptr()), "mavlm_linear_ws")
    return out


def attention(q, k, v, heads, want_lse=False, out=None, head_dim=128, wide_kernel=False, scale=None, plain=False):
    """ctx = softmax(q k^T / sqrt(head_dim)) v per head.  q [R,>=H*hd], k/v [S,...] may be column slices of wider
    buffers.  head_dim 128 (default kernels) or 448 (wide-head kernel; `wide_kernel=True` forces that kernel at 128
    for cross-checks).  `plain` (head_dim 128): the never-split grid of `mavlm_attention` - what the fused step runs for the
    last formation layer when it carries the frame scores (`mavlm_frame_scores_fused`).
    Returns (ctx [R,H*hd], lse2 [H,R] fp32 | None)."""
    _need_gpu(q, k, v)
    R, _, ldq = _rows(q)
    S, _, ldk = _rows(k)
    S2, _, ldv = _rows(v)
    W = heads * head_dim
    if S2 != S or q.shape[1] < W or k.shape[1] < W or v.shape[1] < W or head_dim not in (128, 224, 256, 448):
        raise capi.MavlmError("attention: operand mismatch (head_dim must be 128, 224, 256 or 448)")
    if out is None:
        out = torch.empty((R, W), device=q.device, dtype=q.dtype)
    lse = torch.empty((heads, R), device=q.device, dtype=torch.float32) if want_lse else None
    lp = lse.data_ptr() if want_lse else 0
    if head_dim == 128 and not wide_kernel and plain:
        capi.check(capi.lib().mavlm_attention(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, out.data_ptr(),
                                              out.stride(0), lp, R, S, heads,
                                              1.0 / math.sqrt(128.0) if scale is None else float(scale),
                                              dtype_code(q.dtype), stream_ptr()), "mavlm_attention")
    elif head_dim == 128 and not wide_kernel:
        lib = capi.lib()
        nws = lib.mavlm_attention_ws_floats(R, S, heads)     # > 0: small grid, the keys are split (same plan as mavlm_step)
        ws = torch.empty((nws,), device=q.device, dtype=torch.float32) if nws else None
        capi.check(lib.mavlm_attention_ws(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, out.data_ptr(),
                                          out.stride(0), lp, R, S, heads,
                                          1.0 / math.sqrt(128.0) if scale is None else float(scale),
                                          ws.data_ptr() if nws else 0, nws, dtype_code(q.dtype), stream_ptr()),
                   "mavlm_attention_ws")
    else:
        lib = capi.lib()
        nws = lib.mavlm_attention_hd_ws_floats(R, S, heads, head_dim)    # > 0: small grid, keys split (as mavlm_step)
        ws = torch.empty((nws,), device=q.device, dtype=torch.float32) if nws else None
        capi.check(lib.mavlm_attention_hd_ws(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, out.data_ptr(),
                                             out.stride(0), lp, R, S, heads, head_dim,
                                             1.0 / math.sqrt(head_dim) if scale is None else float(scale),
                                             ws.data_ptr() if nws else 0, nws, dtype_code(q.dtype), stream_ptr()),
                   "mavlm_attention_hd_ws")
    return out, lse


def attention_frames(q, k, v, heads, patches, want_lse=False, scale=None):
    """`attention(..., plain=True)` plus the frame scores in one pass (head_dim 128; the keys are k.shape[0] / patches
    frames of `patches` keys): returns (ctx, lse2 | None, scores [F] fp32)."""
    _need_gpu(q, k, v)
    R, _, ldq = _rows(q)
    S, _, ldk = _rows(k)
    _, _, ldv = _rows(v)
    W = heads * 128
    lib = capi.lib()
    nws = int(lib.mavlm_attention_frames_ws_floats(R, S, heads, patches))
    if nws == 0:
        raise capi.MavlmError("attention_frames: shape not supported (patches % 4, S % patches, <= 64 frames)")
    out = torch.empty((R, W), device=q.device, dtype=q.dtype)
    lse = torch.empty((heads, R), device=q.device, dtype=torch.float32) if want_lse else None
    ws = torch.empty((nws,), device=q.device, dtype=torch.float32)
    scores = torch.empty((S // patches,), device=q.device, dtype=torch.float32)
    capi.check(lib.mavlm_attention_frames(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, out.data_ptr(),
                                          out.stride(0), lse.data_ptr() if want_lse else 0, R, S, heads,
                                          1.0 / math.sqrt(128.0) if scale is None else float(scale), patches,
                                          ws.data_ptr(), nws, scores.data_ptr(), dtype_code(q.dtype), stream_ptr()),
               "mavlm_attention_frames")
    return out, lse, scores


def attention_colsum(q, k, lse2, heads, head_dim=128, wide_kernel=False, scale=None):
    """part[h,s] = sum_q softmax probability of key s for head h (fp32)."""
    _need_gpu(q, k, lse2)
    R, _, ldq = _rows(q)
    S, _, ldk = _rows(k)
    if head_dim == 128 and not wide_kernel:
        # `part` doubles as the pass's scratch (one [H,S] plane per piece of the balanced schedule); result = first plane
        buf = torch.empty((int(capi.lib().mavlm_attention_colsum_floats(R, S, heads)),), device=q.device, dtype=torch.float32)
        part = buf[:heads * S].view(heads, S)
        capi.check(capi.lib().mavlm_attention_colsum(q.data_ptr(), ldq, k.data_ptr(), ldk, lse2.data_ptr(), part.data_ptr(),
                                                     buf.numel(), R, S, heads, 1.0 / math.sqrt(128.0) if scale is None else float(scale),
                                                     dtype_code(q.dtype), stream_ptr()),
                   "mavlm_attention_colsum")
    else:
        part = torch.empty((heads, S), device=q.device, dtype=torch.float32)
        capi.check(capi.lib().mavlm_attention_colsum_hd(q.data_ptr(), ldq, k.data_ptr(), ldk, lse2.data_ptr(),
                                                        part.data_ptr(), R, S, heads, head_dim,
                                                        1.0 / math.sqrt(head_dim) if scale is None else float(scale),
                                                        dtype_code(q.dtype), stream_ptr()), "mavlm_attention_colsum_hd")
    return part


def layernorm(x_f32, gamma_f32, beta_f32, eps, out_dtype, out=None, residual=None):
    """out = LayerNorm(x_f32 + residual) * gamma + beta; residual: optional 16-bit [rows, D] (added in fp32)."""
    _need_gpu(x_f32, gamma_f32, beta_f32, residual)
    if x_f32.dtype != torch.float32 or not x_f32.is_contiguous():
        raise capi.MavlmError("layernorm: fp32 contiguous input expected")
    rows, D = x_f32.shape
    if out is None:
        out = torch.empty((rows, D), device=x_f32.device, dtype=out_dtype)
    rp, ldr = 0, 0
    if residual is not None:
        if residual.dtype != out_dtype or residual.shape != x_f32.shape:
            raise capi.MavlmError("layernorm: residual mismatch")
        _, _, ldr = _rows(residual)
        rp = residual.data_ptr()
    capi.check(capi.lib().mavlm_layernorm(x_f32.data_ptr(), rp, ldr, gamma_f32.data_ptr(), beta_f32.data_ptr(),
                                          out.data_ptr(), rows, D, float(eps), dtype_code(out_dtype), stream_ptr()),
               "mavlm_layernorm")
    return out


_LN_WS = {}          # (device, stream) -> scratch of the fused dense + residual + LayerNorm operator
_LN_WS_KEEP = []     # outgrown scratches (kept alive: captured graphs may still point at them)


def linear_residual_layernorm(x, weight, bias_f32, residual, gamma_f32, beta_f32, eps, want_pre=False, out=None):
    """The Residual block, out = LayerNorm(x @ weight.T + bias + residual) * gamma + beta (MemoryController.py:20-29), the
    way the fused step runs it for this shape: ONE kernel (`mavlm_linear_ln`) where the GEMM fills the chip, else the GEMM
    with fp32 epilogue + the row LayerNorm kernel.  Returns (out 16-bit [M,N], pre fp32 [M,N] = x W^T + bias | None)."""
    _need_gpu(x, weight, bias_f32, residual, gamma_f32, beta_f32)
    M, K, lda = _rows(x)
    N, K2, ldw = _rows(weight)
    _, _, ldr = _rows(residual)
    if K2 != K or weight.dtype != x.dtype or residual.dtype != x.dtype or residual.shape != (M, N):
        raise capi.MavlmError("linear_residual_layernorm: operand mismatch")
    lib = capi.lib()
    nws = int(lib.mavlm_linear_ln_ws_bytes(M, N, K))
    if nws == 0:
        pre = linear(x, weight, bias_f32, capi.EPI_F32)
        return layernorm(pre, gamma_f32, beta_f32, eps, x.dtype, out=out, residual=residual), (pre if want_pre else None)
    # Scratch with a launch counter: ONE per (device, stream) - the kernel's exchange assumes that the launches sharing a
    # scratch are ordered (one stream): two streams on one scratch would read the same epoch and publish into the same
    # granules.  Zero-filled once, then reused; never freed or shrunk (a hipGraph captured around the operator keeps the
    # pointer it was captured with: a larger shape gets a NEW scratch, the old one stays alive in _LN_WS_KEEP).  Never
    # created inside a capture: a replayed memset would reset the launch counter under the previous replay's granules.
    key = (x.device.index, stream_ptr())
    ws = _LN_WS.get(key)
    if ws is None or ws.numel() < nws:
        if torch.cuda.is_current_stream_capturing():
            raise capi.MavlmError("linear_residual_layernorm: first use inside a graph capture - warm the operator up "
                                  "outside the capture ON THE STREAM THAT CAPTURES (torch.cuda.graph(g, stream=warm_stream)): its scratch carries "
                                  "a launch counter and there is one per (device, stream)")
        if ws is not None:
            _LN_WS_KEEP.append(ws)
        ws = _LN_WS[key] = torch.zeros(max(nws, 1 << 23), device=x.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=x.dtype)
    pre = torch.empty((M, N), device=x.device, dtype=torch.float32) if want_pre else None
    capi.check(lib.mavlm_linear_ln(x.data_ptr(), lda, weight.data_ptr(), ldw, bias_f32.data_ptr(), residual.data_ptr(), ldr,
                                   gamma_f32.data_ptr(), beta_f32.data_ptr(), float(eps), out.data_ptr(), out.stride(0),
                                   pre.data_ptr() if want_pre else 0, M, N, K, ws.data_ptr(), ws.numel(),
                                   dtype_code(x.dtype), stream_ptr()), "mavlm_linear_ln")
    return out, pre


def row_add(x, table, idx=None, src=None, out=None):
    """out[t,p,:] = x[src[t],p,:] + table[idx[t],:]  (x [T0,P,D]; idx/src int64 device tensors or None)."""
    _need_gpu(x, table, idx, src, out)
    if x.dim() != 3 or not x.is_contiguous() or not table.is_contiguous() or table.dtype != x.dtype:
        raise capi.MavlmError("row_add: operand mismatch")
    T = x.shape[0] if src is None else src.numel()
    if idx is not None and idx.numel() != T:
        raise capi.MavlmError("row_add: index length mismatch")
    P, D = x.shape[1], x.shape[2]
    if out is None:
        out = torch.empty((T, P, D), device=x.device, dtype=x.dtype)
    capi.check(capi.lib().mavlm_row_add(x.data_ptr(), src.data_ptr() if src is not None else 0, table.data_ptr(),
                                        idx.data_ptr() if idx is not None else 0, out.data_ptr(), T, P, D,
                                        dtype_code(x.dtype), stream_ptr()), "mavlm_row_add")
    return out


def pool_bilinear(x, side, stride=2, pe_table=None, idx=None, out=None):
    """[F, side*side, D] -> [F, ceil(side/stride)**2, D], bilinear (align_corners=False); optional fused PE add."""
    _need_gpu(x, pe_table, idx, out)
    if x.dim() != 3 or not x.is_contiguous() or x.shape[1] != side * side:
        raise capi.MavlmError("pool_bilinear: expected contiguous [F, side*side, D]")
    F, _, D = x.shape
    os_ = -(-side // stride)
    if out is None:
        out = torch.empty((F, os_ * os_, D), device=x.device, dtype=x.dtype)
    if pe_table is not None and (idx is None or idx.numel() != F or pe_table.dtype != x.dtype):
        raise capi.MavlmError("pool_bilinear: PE table / index mismatch")
    capi.check(capi.lib().mavlm_pool_bilinear(x.data_ptr(), out.data_ptr(), pe_table.data_ptr() if pe_table is not None else 0,
                                              idx.data_ptr() if idx is not None else 0, F, side, stride, D,
                                              dtype_code(x.dtype), stream_ptr()), "mavlm_pool_bilinear")
    return out


# ---- backward-pass ops -----------------------------------------------------------------------
def attention_bwd(q, k, v, o, do, lse2, heads, need_dq=True, need_dk=True, need_dv=True, scale=None):
    """Gradients of `attention` (head_dim 128): returns (dq | None, dk | None, dv | None), each [rows, H*128]."""
    _need_gpu(q, k, v, o, do, lse2)
    R, _, ldq = _rows(q)
    S, _, ldk = _rows(k)
    _, _, ldv = _rows(v)
    _, _, ldo = _rows(o)
    _, _, lddo = _rows(do)
    W = heads * 128
    if o.shape != (R, W) or do.shape != (R, W) or lse2.shape != (heads, R) or lse2.dtype != torch.float32 \
            or not lse2.is_contiguous() or v.shape[0] != S or do.dtype != q.dtype:
        raise capi.MavlmError("attention_bwd: operand mismatch")
    dq = torch.empty((R, W), device=q.device, dtype=q.dtype) if need_dq else None
    dk = torch.empty((S, W), device=q.device, dtype=q.dtype) if need_dk else None
    dv = torch.empty((S, W), device=q.device, dtype=q.dtype) if need_dv else None
    delta = torch.empty((heads, R), device=q.device, dtype=torch.float32)
    p = lambda t: t.data_ptr() if t is not None else 0
    capi.check(capi.lib().mavlm_attention_bwd(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, o.data_ptr(), ldo,
                                              do.data_ptr(), lddo, lse2.data_ptr(), delta.data_ptr(), p(dq), W, p(dk), W,
                                              p(dv), W, R, S, heads,
                                              1.0 / math.sqrt(128.0) if scale is None else float(scale),
                                              dtype_code(q.dtype), stream_ptr()), "mavlm_attention_bwd")
    return dq, dk, dv


def transpose(x, pad_to=64, out_rows=None):
    """[rows, cols] 16-bit -> [cols, roundup(rows, 64)] with a zero-filled pad (the K-contiguous operand form of
    the contract-over-rows products).  `out_rows` > cols: the result gets that many rows, the extra ones zero (an
    operand whose row count must be a multiple of 128 for the GEMM, e.g. a 448-wide head -> 512)."""
    _need_gpu(x)
    rows, cols, ld = _rows(x)
    rp = -(-rows // pad_to) * pad_to
    if out_rows is None or out_rows == cols:
        out = torch.empty((cols, rp), device=x.device, dtype=x.dtype)
    else:
        out = torch.zeros((out_rows, rp), device=x.device, dtype=x.dtype)
    capi.check(capi.lib().mavlm_transpose(x.data_ptr(), ld, rows, cols, out.data_ptr(), rp, stream_ptr()),
               "mavlm_transpose")
    return out


def rowsum(x, cols=None):
    """fp32 row sums of a 16-bit [rows, >=cols] tensor."""
    _need_gpu(x)
    rows, c, ld = _rows(x)
    out = torch.empty((rows,), device=x.device, dtype=torch.float32)
    capi.check(capi.lib().mavlm_rowsum(x.data_ptr(), ld, rows, cols if cols is not None else c, out.data_ptr(),
                                       dtype_code(x.dtype), stream_ptr()), "mavlm_rowsum")
    return out


_ZERO_BIAS = {}


def zero_bias(n, device):
    z = _ZERO_BIAS.get((n, device))
    if z is None:
        z = _ZERO_BIAS[(n, device)] = torch.zeros(n, device=device, dtype=torch.float32)
    return z


def matmul_nt(a, b):
    """a [M,K] . b [N,K]^T -> [M,N] 16-bit (no bias)."""
    return linear(a, b, zero_bias(b.shape[0], a.device))


def matmul_nt_splitk(a, b, splits=None):
    """As matmul_nt for long contractions with few output tiles (dW = dY^T X): split-K with fp32 partials."""
    _need_gpu(a, b)
    M, K, lda = _rows(a)
    N, K2, ldb = _rows(b)
    if K2 != K or a.dtype != b.dtype:
        raise capi.MavlmError("matmul_nt_splitk: operand mismatch")
    tiles = -(-M // 128) * (N // 128)
    if splits is None:
        splits = max(1, min(K // 64, -(-512 // tiles)))
    if splits == 1:
        # enough output tiles to fill the chip (every dW of the OneVision-7B width): the plain GEMM with its automatic tile
        # choice (256-row kernels), 16-bit output written directly - no fp32 plane, no reduction pass.  Same bits: every kernel
        # sums an element's K products in the same order.
        return matmul_nt(a, b)
    out = torch.empty((M, N), device=a.device, dtype=a.dtype)
    ws = torch.empty((splits, M, N), device=a.device, dtype=torch.float32)
    capi.check(capi.lib().mavlm_linear_splitk(a.data_ptr(), lda, b.data_ptr(), ldb, out.data_ptr(), M, N, K, splits,
                                              ws.data_ptr(), zero_bias(N, a.device).data_ptr(), dtype_code(a.dtype),
                                              stream_ptr()), "mavlm_linear_splitk")
    return out


def layernorm_bwd(dy, x_f32, residual, gamma_f32, eps):
    """Backward of `layernorm`: returns (dz [rows,D] 16-bit = grad of x and of residual, dgamma fp32, dbeta fp32)."""
    _need_gpu(dy, x_f32, residual, gamma_f32)
    rows, D = x_f32.shape
    if dy.shape != x_f32.shape or not dy.is_contiguous() or not x_f32.is_contiguous() or x_f32.dtype != torch.float32:
        raise capi.MavlmError("layernorm_bwd: operand mismatch")
    dz = torch.empty((rows, D), device=dy.device, dtype=dy.dtype)
    dg = torch.empty((D,), device=dy.device, dtype=torch.float32)
    db = torch.empty((D,), device=dy.device, dtype=torch.float32)
    ws = torch.empty((capi.lib().mavlm_layernorm_bwd_ws_floats(D),), device=dy.device, dtype=torch.float32)
    rp, ldr = 0, 0
    if residual is not None:
        _, _, ldr = _rows(residual)
        rp = residual.data_ptr()
    capi.check(capi.lib().mavlm_layernorm_bwd(dy.data_ptr(), x_f32.data_ptr(), rp, ldr, gamma_f32.data_ptr(), dz.data_ptr(),
                                              dg.data_ptr(), db.data_ptr(), ws.data_ptr(), rows, D, float(eps),
                                              dtype_code(dy.dtype), stream_ptr()), "mavlm_layernorm_bwd")
    return dz, dg, db


ACT_GELU, ACT_GELU_BWD, ACT_RELU_BWD = 0, 1, 2


def act(kind, x, dy=None):
    _need_gpu(x, dy)
    if not x.is_contiguous() or (dy is not None and (not dy.is_contiguous() or dy.shape != x.shape)):
        raise capi.MavlmError("act: contiguous operands of one shape expected")
    out = torch.empty_like(x)
    capi.check(capi.lib().mavlm_act(kind, x.data_ptr(), dy.data_ptr() if dy is not None else 0, out.data_ptr(), x.numel(),
                                    dtype_code(x.dtype), stream_ptr()), "mavlm_act")
    return out


# ---- inactive variants ----------------------------------------------------------------------
def frame_mean(x, want_f32=False):
    """[F,P,D] 16-bit -> per-frame mean over the patches: [F,D] in x.dtype (and fp32 when asked)."""
    _need_gpu(x)
    if x.dim() != 3 or not x.is_contiguous():
        raise capi.MavlmError("frame_mean: contiguous [F,P,D] expected")
    F, P, D = x.shape
    o16 = torch.empty((F, D), device=x.device, dtype=x.dtype)
    o32 = torch.empty((F, D), device=x.device, dtype=torch.float32) if want_f32 else None
    capi.check(capi.lib().mavlm_frame_mean(x.data_ptr(), o16.data_ptr(), o32.data_ptr() if want_f32 else 0, F, P, D,
                                           dtype_code(x.dtype), stream_ptr()), "mavlm_frame_mean")
    return (o16, o32) if want_f32 else o16


def adjacent_cosine(v_f32, eps):
    """cosine_similarity(v[:-1], v[1:], eps) for fp32 rows -> fp32 [n-1]."""
    _need_gpu(v_f32)
    if v_f32.dim() != 2 or v_f32.dtype != torch.float32 or not v_f32.is_contiguous() or v_f32.shape[0] < 2:
        raise capi.MavlmError("adjacent_cosine: contiguous fp32 [n>=2, D] expected")
    n, D = v_f32.shape
    out = torch.empty((n - 1,), device=v_f32.device, dtype=torch.float32)
    capi.check(capi.lib().mavlm_adjacent_cosine(v_f32.data_ptr(), out.data_ptr(), n, D, float(eps), stream_ptr()),
               "mavlm_adjacent_cosine")
    return out


def gru_sequence(xg_f32, whh, bhh_f32, hidden, ndir):
    """Recurrent half of nn.GRU: xg fp32 [F, ndir*3H], whh [ndir,3H,H] 16-bit, bhh fp32 [ndir,3H] -> [F, ndir*H]."""
    _need_gpu(xg_f32, whh, bhh_f32)
    F = xg_f32.shape[0]
    if xg_f32.shape != (F, ndir * 3 * hidden) or tuple(whh.shape) != (ndir, 3 * hidden, hidden) or \
            tuple(bhh_f32.shape) != (ndir, 3 * hidden) or not (xg_f32.is_contiguous() and whh.is_contiguous()
                                                              and bhh_f32.is_contiguous()):
        raise capi.MavlmError("gru_sequence: operand mismatch")
    out = torch.empty((F, ndir * hidden), device=whh.device, dtype=whh.dtype)
    capi.check(capi.lib().mavlm_gru_sequence(xg_f32.data_ptr(), whh.data_ptr(), bhh_f32.data_ptr(), out.data_ptr(), F,
                                             hidden, ndir, dtype_code(whh.dtype), stream_ptr()), "mavlm_gru_sequence")
    return out


WIDE_BWD_STREAMS = 4        # heads of the GEMM-composed wide-head backward in flight (1 = rounds 1-3: one head after the other)
_WIDE_BWD_POOL = {}


def _wide_bwd_streams(dev, n):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _WIDE_BWD_POOL:
        _WIDE_BWD_POOL[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _WIDE_BWD_POOL[key]


WIDE_BWD_FLASH = True       # head_dim 448: the flash-style kernels of attention_bwd_hd.hip (False: the GEMM-composed form below)


def attention_bwd_hd(q, k, v, o, do, lse2, heads, head_dim, scale, need_dq=True, need_dk=True, need_dv=True):
    """Gradients of `attention` at head_dim 448 (LLaVA-OneVision-7B), flash style (`mavlm_attention_bwd_hd`): the
    probabilities are recomputed per tile from lse2, nothing of size [R, S] is materialised.  Returns (dq, dk, dv)."""
    _need_gpu(q, k, v, o, do, lse2)
    R, _, ldq = _rows(q)
    S, _, ldk = _rows(k)
    _, _, ldv = _rows(v)
    _, _, ldo = _rows(o)
    do = do if do.stride(-1) == 1 else do.contiguous()
    _, _, lddo = _rows(do)
    W = heads * head_dim
    if o.shape != (R, W) or do.shape != (R, W) or lse2.shape != (heads, R) or lse2.dtype != torch.float32 \
            or not lse2.is_contiguous() or v.shape[0] != S or do.dtype != q.dtype:
        raise capi.MavlmError("attention_bwd_hd: operand mismatch")
    dq = torch.empty((R, W), device=q.device, dtype=q.dtype) if need_dq else None
    dk = torch.empty((S, W), device=q.device, dtype=q.dtype) if need_dk else None
    dv = torch.empty((S, W), device=q.device, dtype=q.dtype) if need_dv else None
    delta = torch.empty((heads, R), device=q.device, dtype=torch.float32)
    p = lambda t: t.data_ptr() if t is not None else 0
    capi.check(capi.lib().mavlm_attention_bwd_hd(q.data_ptr(), ldq, k.data_ptr(), ldk, v.data_ptr(), ldv, o.data_ptr(), ldo,
                                                 do.data_ptr(), lddo, lse2.data_ptr(), delta.data_ptr(), p(dq), W, p(dk), W,
                                                 p(dv), W, R, S, heads, head_dim, float(scale), dtype_code(q.dtype),
                                                 stream_ptr()), "mavlm_attention_bwd_hd")
    return dq, dk, dv


def attention_bwd_wide(q, k, v, o, do, lse2, heads, head_dim, scale, need_dq=True, need_dk=True, need_dv=True):
    """Gradients of `attention` for wide heads.  head_dim 448 runs the flash-style kernels (`attention_bwd_hd`); the form
    below - rounds 2-4, kept for the other multiples of 64 and as the cross-check of the kernels - composes the backward from
    GEMMs: ONE head's [R,S] scores are materialised at a
    time (fp32, from the MFMA GEMM) and every product of the backward runs as a GEMM of this library:
        S = q_h k_h^T  ->  P = exp2(S c - lse2)  ->  dP = dO_h v_h^T  ->  dS = P o (dP - delta) scale
        dV_h = P^T dO_h,   dQ_h = dS k_h,   dK_h = dS^T q_h
    (same rounding points as the flash-style path: P and dS in 16 bits, everything else fp32 accumulation)."""
    if head_dim == 448 and WIDE_BWD_FLASH:
        return attention_bwd_hd(q, k, v, o, do, lse2, heads, head_dim, scale, need_dq, need_dk, need_dv)
    _need_gpu(q, k, v, o, do, lse2)
    R, S = q.shape[0], k.shape[0]
    hd, H = head_dim, heads
    W = H * hd
    if o.shape != (R, W) or do.shape != (R, W) or lse2.shape != (H, R) or hd % 64 or v.shape[0] != S:
        raise capi.MavlmError("attention_bwd_wide: operand mismatch")
    lib, dt, dev = capi.lib(), q.dtype, q.device
    code = dtype_code(dt)
    hp = -(-hd // 128) * 128                       # output width the GEMM can produce (448 -> 512)
    Sp = -(-S // 128) * 128
    delta = torch.empty((H, R), device=dev, dtype=torch.float32)
    do = do if do.is_contiguous() else do.contiguous()
    capi.check(lib.mavlm_rowdot_heads(do.data_ptr(), do.stride(0), o.data_ptr(), o.stride(0), delta.data_ptr(), R, H, hd,
                                      code, stream_ptr()), "mavlm_rowdot_heads")
    dq = torch.empty((R, W), device=dev, dtype=dt) if need_dq else None
    dk = torch.empty((S, W), device=dev, dtype=dt) if need_dk else None
    dv = torch.empty((S, W), device=dev, dtype=dt) if need_dv else None

    def rows_padded(t):                            # [S, hd] view -> [Sp, hd] with zero rows (GEMM N % 128)
        if Sp == S:
            return t
        out = torch.zeros((Sp, hd), device=dev, dtype=dt)
        out[:S] = t
        return out

    # Round 4: the heads are independent chains of small, short kernels (K = 448: 7 K-tiles per GEMM, 1-1.3 rounds of
    # tiles, ~30 launches per head): one after the other they leave most of the chip idle between and inside launches.  Head h
    # runs on side stream h % WIDE_BWD_STREAMS; the streams fork from the current one and join it at the end.
    cur = torch.cuda.current_stream()
    zero_bias(Sp, dev), zero_bias(hp, dev)                    # (cached constants: created on the current stream, before the fork)
    ns = max(1, min(WIDE_BWD_STREAMS, H))
    pool = _wide_bwd_streams(dev, ns) if ns > 1 and not torch.cuda.is_current_stream_capturing() else [cur]
    for st in pool:
        if st is not cur:
            st.wait_stream(cur)
    for h in range(H):
      with torch.cuda.stream(pool[h % len(pool)]):
          sl = slice(h * hd, (h + 1) * hd)
          qh, doh = q[:, sl], do[:, sl]
          kc, vc = rows_padded(k[:, sl]), rows_padded(v[:, sl])
          s32 = linear(qh, kc, zero_bias(Sp, dev), capi.EPI_F32)                       # [R, Sp] raw scores
          p16 = torch.empty((R, Sp), device=dev, dtype=dt)
          capi.check(lib.mavlm_attention_probs(s32.data_ptr(), Sp, lse2[h].data_ptr(), p16.data_ptr(), Sp, R, Sp, S,
                                               float(scale), code, stream_ptr()), "mavlm_attention_probs")
          if need_dv:
              dvh = linear(transpose(p16), transpose(doh, out_rows=hp), zero_bias(hp, dev))        # [Sp, hp]
              dv[:, sl] = dvh[:S, :hd]
          if need_dq or need_dk:
              dp32 = linear(doh, vc, zero_bias(Sp, dev), capi.EPI_F32)
              ds16 = torch.empty((R, Sp), device=dev, dtype=dt)      # from the UNROUNDED probability, as the flash kernels
              capi.check(lib.mavlm_attention_dscores(s32.data_ptr(), Sp, dp32.data_ptr(), Sp, lse2[h].data_ptr(),
                                                     delta[h].data_ptr(), ds16.data_ptr(), Sp, R, Sp, S, float(scale), code,
                                                     stream_ptr()), "mavlm_attention_dscores")
              if need_dq:
                  dqh = linear(ds16, transpose(kc, out_rows=hp), zero_bias(hp, dev))               # [R, hp]
                  dq[:, sl] = dqh[:, :hd]
              if need_dk:
                  dkh = linear(transpose(ds16), transpose(qh, out_rows=hp), zero_bias(hp, dev))    # [Sp, hp]
                  dk[:, sl] = dkh[:S, :hd]
    for st in pool:
        if st is not cur:
            cur.wait_stream(st)
    if len(pool) > 1:                                          # (tensors of the current stream used on the side streams)
        for t in (dq, dk, dv, q, k, v, do, lse2, delta):
            if t is not None:
                for st in pool:
                    t.record_stream(st)
    return dq, dk, dv
