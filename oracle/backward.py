"""CPU oracle for the BACKWARD of the memory path's operators (test infrastructure only - never imported by the
product; see oracle/memory_path.py for the rules).

The reference has no hand-written backward: it differentiates `llava/model/memory_module/MemoryController.py:48-71`
and `llava/model/llava_arch.py:132-136,546-554` with torch autograd.  The functions below restate the closed-form
gradients of those same expressions in numpy, with the 16-bit rounding points of the HIP kernels
(`csrc/attention_bwd.hip`, `csrc/backward.hip`) emulated so that operator-level parity can be gated tightly.
They are pinned in `tests/test_oracle_backward.py` against torch autograd (fp64) of the reference expressions and,
through `oracle/torch_path.py`, against the gradients of the imported reference modules (tests/golden/g8_grads.npz).
"""
import math

import numpy as np

from . import memory_path as O

F32 = np.float32


def attention_bwd(Q, K, V, Oc, dO, lse2, heads, mode="fp32", scale_in_ds=False):
    """Gradients of ctx = softmax(Q K^T / sqrt(d)) V per head (MemoryController.py:51-54).

    Oc = forward output (as stored), lse2 = [H,R] log2-domain log-sum-exp of the forward.  Rounding points of the
    kernel (emulation modes): P rounded to 16 bits as the operand of dV = P^T dO; dS = P o (dP - delta) rounded as
    the operand of dQ = dS K and dK = dS^T Q; delta = sum_d dO*O in fp32; the 1/sqrt(d) factor applied to the fp32
    accumulators.  `scale_in_ds`: the wide-head path (ops.attention_bwd_wide) folds 1/sqrt(d) into dS BEFORE its
    16-bit rounding (its products are plain GEMMs without an output scale) - same accuracy, different rounding point.
    Returns (dQ, dK, dV) unrounded float32."""
    r = O.rounder(mode)
    R, S = Q.shape[0], K.shape[0]
    d = Q.shape[1] // heads
    scale = F32(1.0 / math.sqrt(d))
    c = F32(scale * F32(1.4426950408889634))
    dQ = np.empty((R, heads * d), F32)
    dK = np.empty((S, heads * d), F32)
    dV = np.empty((S, heads * d), F32)
    for h in range(heads):
        sl = slice(h * d, (h + 1) * d)
        s = O._mm(Q[:, sl], K[:, sl].T)
        p = np.exp2(s * c - lse2[h].reshape(R, 1).astype(F32), dtype=F32)
        delta = (dO[:, sl].astype(F32) * Oc[:, sl].astype(F32)).sum(axis=1, keepdims=True, dtype=F32)
        dp = O._mm(dO[:, sl], V[:, sl].T)
        ds = r(p * (dp - delta) * scale) if scale_in_ds else r(p * (dp - delta))
        post = F32(1.0) if scale_in_ds else scale
        dV[:, sl] = O._mm(r(p).T, dO[:, sl])
        dQ[:, sl] = O._mm(ds, K[:, sl]) * post
        dK[:, sl] = O._mm(ds.T, Q[:, sl]) * post
    return dQ, dK, dV


def layernorm_bwd(dy, x, res, gamma, eps):
    """Backward of y = LN(x + res)*gamma + beta (MemoryController.py:24,28; biased variance).
    Returns (dz = grad of x and of res, dgamma, dbeta), float32 from float64 arithmetic."""
    z = x.astype(np.float64) + (0.0 if res is None else res.astype(np.float64))
    mu = z.mean(axis=1, keepdims=True)
    var = ((z - mu) ** 2).mean(axis=1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = (z - mu) * rstd
    dyf = dy.astype(np.float64)
    g = dyf * gamma.astype(np.float64)
    dz = rstd * (g - g.mean(axis=1, keepdims=True) - xh * (g * xh).mean(axis=1, keepdims=True))
    return dz.astype(F32), (dyf * xh).sum(axis=0).astype(F32), dyf.sum(axis=0).astype(F32)


def linear_bwd(dY, X, W):
    """y = x W^T + b (nn.Linear): dX = dY W, dW = dY^T X, db = column sums of dY.  Unrounded float32."""
    return O._mm(dY, W), O._mm(dY.T, X), dY.astype(F32).sum(axis=0, dtype=F32)


def gelu_bwd(x, dy):
    from scipy.special import erf
    xf = x.astype(np.float64)
    cdf = 0.5 * (1.0 + erf(xf / math.sqrt(2.0)))
    pdf = np.exp(-0.5 * xf * xf) / math.sqrt(2.0 * math.pi)
    return (dy.astype(np.float64) * (cdf + xf * pdf)).astype(F32)


def relu_bwd(y, dy):
    return np.where(y > 0, dy, 0).astype(F32)
