"""-m gpu: operator-level parity of the BACKWARD kernels (through the C ABI) against oracle/backward.py.

Gates: exact for layout ops and integer-valued data; rel-L2 <= 1e-3 against the oracle with the kernels' 16-bit
rounding points emulated (attention: P and dS rounded as MFMA operands)."""
import numpy as np
import pytest
import torch

import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi
from memory_augmented_vlm_amd import _ops as ops
from oracle import memory_path as O
from oracle import backward as OB
from gpu_util import to_dev, to_np, f32_dev

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _int_mat(shape, seed, lo=-4, hi=4):
    return np.floor(O.hash_uniform(shape, seed, lo, hi + 0.999)).astype(np.float32)


@pytest.mark.parametrize("rows,cols", [(64, 64), (1, 128), (130, 64), (1568, 1024), (200, 4096), (6272, 256)])
def test_transpose_exact(rows, cols):
    a = _int_mat((rows, cols), 3, -100, 100)
    x = to_dev(a, "bf16")
    t = ops.transpose(x)
    rp = -(-rows // 64) * 64
    assert tuple(t.shape) == (cols, rp)
    got = to_np(t)
    assert np.array_equal(got[:, :rows], a.T)
    assert not got[:, rows:].any()
    # a column slice of a wider buffer (ld > cols) as input
    if cols >= 128:
        t2 = ops.transpose(x[:, 64:128])
        assert np.array_equal(to_np(t2)[:, :rows], a[:, 64:128].T)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_rowsum_and_splitk(mode):
    a = _int_mat((256, 1000), 5, -3, 3)
    x = to_dev(np.pad(a, ((0, 0), (0, 24))), mode)
    assert np.array_equal(to_np(ops.rowsum(x, 1000)), a.sum(axis=1))
    # dW-shaped product: short outputs, long contraction; integer data -> exact for every split count
    A = _int_mat((256, 4160), 6, -2, 2)
    B = _int_mat((128, 4160), 7, -2, 2)
    ref = A @ B.T
    for splits in (1, 3, 8, None):
        got = to_np(ops.matmul_nt_splitk(to_dev(A, mode), to_dev(B, mode), splits))
        assert np.array_equal(got, O.rounder(mode)(ref)), splits
    got = to_np(ops.matmul_nt(to_dev(A, mode), to_dev(B, mode)))
    assert np.array_equal(got, O.rounder(mode)(ref))


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("rows,D", [(5, 128), (300, 1024), (1568, 896), (70, 3584)])
def test_layernorm_bwd(mode, rows, D):
    r = O.rounder(mode)
    x = O.hash_normal_like((rows, D), 11, 1.5).astype(np.float32)
    res = r(O.hash_normal_like((rows, D), 12, 1.0))
    dy = r(O.hash_normal_like((rows, D), 13, 0.7))
    g = (1.0 + 0.2 * O.hash_normal_like((D,), 14)).astype(np.float32)
    for with_res in (True, False):
        dz, dg, db = ops.layernorm_bwd(to_dev(dy, mode), f32_dev(x), to_dev(res, mode) if with_res else None, f32_dev(g),
                                       1e-12)
        rz, rg, rb = OB.layernorm_bwd(dy, x, res if with_res else None, g, 1e-12)
        assert O.rel_l2(to_np(dz), r(rz)) < TOL
        assert O.rel_l2(to_np(dg), rg) < 1e-4
        assert O.rel_l2(to_np(db), rb) < 1e-4


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_activation_kernels(mode):
    r = O.rounder(mode)
    x = r(O.hash_normal_like((333, 256), 21, 2.0))
    dy = r(O.hash_normal_like((333, 256), 22, 1.0))
    assert O.rel_l2(to_np(ops.act(ops.ACT_GELU, to_dev(x, mode))), r(O.gelu_erf(x))) < 1e-4
    assert O.rel_l2(to_np(ops.act(ops.ACT_GELU_BWD, to_dev(x, mode), to_dev(dy, mode))), r(OB.gelu_bwd(x, dy))) < 1e-4
    y = np.maximum(x, 0)
    assert np.array_equal(to_np(ops.act(ops.ACT_RELU_BWD, to_dev(y, mode), to_dev(dy, mode))), OB.relu_bwd(y, dy))


@pytest.fixture(params=[0, 1])
def bwd_fused(request):
    """Attention-backward tests run with separate dK / dV kernels and with the fused dK+dV kernel."""
    capi.check(capi.lib().mavlm_set_attention_bwd_fused(request.param), "set fused")
    yield request.param
    capi.lib().mavlm_set_attention_bwd_fused(0)


def _attn_case(R, S, H, mode, seed, qs=1.0):
    r = O.rounder(mode)
    W = H * 128
    Q = r(O.hash_normal_like((R, W), seed, qs))
    K = r(O.hash_normal_like((S, W), seed + 1, 1.0))
    V = r(O.hash_normal_like((S, W), seed + 2, 1.0))
    dO = r(O.hash_normal_like((R, W), seed + 3, 0.5))
    return Q, K, V, dO


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("R,S,H", [(128, 64, 1), (32, 200, 2), (196, 392, 8), (300, 130, 2), (1, 2, 1), (129, 65, 1)])
def test_attention_bwd_vs_oracle(mode, R, S, H, bwd_fused):
    r = O.rounder(mode)
    Q, K, V, dO = _attn_case(R, S, H, mode, 31)
    q, k, v, do = (to_dev(a, mode) for a in (Q, K, V, dO))
    o, lse = ops.attention(q, k, v, H, want_lse=True)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, H)
    rq, rk, rv = OB.attention_bwd(Q, K, V, to_np(o), dO, to_np(lse), H, mode)
    assert O.rel_l2(to_np(dv), r(rv)) < TOL
    assert O.rel_l2(to_np(dq), r(rq)) < TOL
    assert O.rel_l2(to_np(dk), r(rk)) < TOL
    # skipping outputs leaves the others unchanged
    dq2, dk2, dv2 = ops.attention_bwd(q, k, v, o, do, lse, H, need_dq=False, need_dk=True, need_dv=False)
    assert dq2 is None and dv2 is None and torch.equal(dk2, dk)


def test_attention_bwd_matches_autograd_fp32(bwd_fused):
    """End check against torch autograd of the reference expression (MemoryController.py:51-54) in fp32 on the
    same 16-bit inputs; the distance is the kernels' operand rounding (P, dS to bf16): a few 1e-3."""
    R, S, H = 260, 330, 2
    Q, K, V, dO = _attn_case(R, S, H, "bf16", 41, 0.5)
    q, k, v, do = (to_dev(a, "bf16") for a in (Q, K, V, dO))
    o, lse = ops.attention(q, k, v, H, want_lse=True)
    dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, H)
    tq, tk, tv = (torch.from_numpy(a).cuda().requires_grad_() for a in (Q, K, V))
    def heads(t):
        return t.view(t.shape[0], H, 128).permute(1, 0, 2)
    s = torch.matmul(heads(tq), heads(tk).transpose(-1, -2)) / 128 ** 0.5
    ctx = torch.matmul(torch.softmax(s, dim=-1), heads(tv)).permute(1, 0, 2).reshape(R, H * 128)
    ctx.backward(torch.from_numpy(dO).cuda())
    for got, ref in ((dq, tq.grad), (dk, tk.grad), (dv, tv.grad)):
        assert O.rel_l2(to_np(got), to_np(ref)) < 8e-3


def test_attention_bwd_strided_operands_and_bad_args(bwd_fused):
    """K and V as column slices of a packed [S, 2W] projection output (as the path stores them); argument checks."""
    R, S, H = 100, 150, 2
    W = H * 128
    Q, K, V, dO = _attn_case(R, S, H, "bf16", 51)
    kv = to_dev(np.concatenate([K, V], axis=1), "bf16")
    q, do = to_dev(Q, "bf16"), to_dev(dO, "bf16")
    o, lse = ops.attention(q, kv[:, :W], kv[:, W:], H, want_lse=True)
    a = ops.attention_bwd(q, kv[:, :W], kv[:, W:], o, do, lse, H)
    b = ops.attention_bwd(q, kv[:, :W].contiguous(), kv[:, W:].contiguous(), o, do, lse, H)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    with pytest.raises(capi.MavlmError):
        ops.attention_bwd(q, kv[:, :W], kv[:, W:], o, do[:50], lse, H)
    with pytest.raises(capi.MavlmError):
        ops.attention_bwd(q.cpu(), kv[:, :W], kv[:, W:], o, do, lse, H)


@pytest.fixture(params=[True, False], ids=["flash", "composed"])
def wide_flash(request):
    """head_dim-448 backward tests run the flash-style kernels (the product) and the GEMM-composed form they replaced."""
    prev = ops.WIDE_BWD_FLASH
    ops.WIDE_BWD_FLASH = request.param
    yield request.param
    ops.WIDE_BWD_FLASH = prev


def _wide_case(R, S, H, mode, seed=61):
    r = O.rounder(mode)
    W = H * 448
    Q = r(O.hash_normal_like((R, W), seed, 0.5))
    K = r(O.hash_normal_like((S, W), seed + 1, 0.5))
    V = r(O.hash_normal_like((S, W), seed + 2, 1.0))
    dO = r(O.hash_normal_like((R, W), seed + 3, 0.5))
    return Q, K, V, dO


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("R,S,H", [(196, 392, 2), (130, 200, 1), (64, 128, 8), (1, 2, 1), (33, 97, 2), (300, 131, 2)])
def test_attention_bwd_wide_heads_vs_oracle(mode, R, S, H, wide_flash):
    """head_dim 448 (LLaVA-OneVision-7B).  flash: `attn_bwd_hd_kernel` (P recomputed per tile from lse2; dS rounded to 16 bits,
    the scale applied to the fp32 result - as the 128-wide kernels); composed: per-head materialised scores, every product a
    GEMM (the scale inside dS).  Same gates as the 128-wide backward; ragged row counts on both sides of every tile size."""
    hd = 448
    if not wide_flash and (R, S) in ((1, 2), (33, 97), (300, 131)):
        pytest.skip("ragged cases: the flash kernels' tile edges")
    r = O.rounder(mode)
    Q, K, V, dO = _wide_case(R, S, H, mode)
    q, k, v, do = (to_dev(a, mode) for a in (Q, K, V, dO))
    o, lse = ops.attention(q, k, v, H, want_lse=True, head_dim=hd)
    scale = ops.attn_scale(hd)
    dq, dk, dv = ops.attention_bwd_wide(q, k, v, o, do, lse, H, hd, scale)
    rq, rk, rv = OB.attention_bwd(Q, K, V, to_np(o), dO, to_np(lse), H, mode, scale_in_ds=not wide_flash)
    assert O.rel_l2(to_np(dv), r(rv)) < TOL
    assert O.rel_l2(to_np(dq), r(rq)) < TOL
    assert O.rel_l2(to_np(dk), r(rk)) < TOL
    a = ops.attention_bwd_wide(q, k, v, o, do, lse, H, hd, scale, need_dq=False, need_dk=True, need_dv=False)
    assert a[0] is None and a[2] is None and torch.equal(a[1], dk)
    b = ops.attention_bwd_wide(q, k, v, o, do, lse, H, hd, scale, need_dq=False, need_dk=False, need_dv=True)
    assert b[0] is None and b[1] is None and torch.equal(b[2], dv)      # (flash: the dV-only kernel against the fused dK + dV one)


def test_attention_bwd_wide_flash_vs_composed_and_strided_operands():
    """The flash-style 448 kernels against the GEMM-composed form on the same inputs (two independent implementations of the
    same rounding points up to where the scale enters), K / V as column slices of a packed [S, 2W] projection output (as the
    path stores them), a path-sized row count (M = 8: 1568 memory rows against a 32-frame chunk + one memory), determinism."""
    hd, H, R, S = 448, 8, 1568, 1568 + 2 * 196
    W = H * hd
    Q, K, V, dO = _wide_case(R, S, H, "bf16", 71)
    kv = to_dev(np.concatenate([K, V], axis=1), "bf16")
    q, do = to_dev(Q, "bf16"), to_dev(dO, "bf16")
    k, v = kv[:, :W], kv[:, W:]
    o, lse = ops.attention(q, k, v, H, want_lse=True, head_dim=hd)
    scale = ops.attn_scale(hd)
    a = ops.attention_bwd_hd(q, k, v, o, do, lse, H, hd, scale)
    b = ops.attention_bwd_hd(q, k.contiguous(), v.contiguous(), o, do, lse, H, hd, scale)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    prev, ops.WIDE_BWD_FLASH = ops.WIDE_BWD_FLASH, False
    try:
        cmp = ops.attention_bwd_wide(q, k.contiguous(), v.contiguous(), o, do, lse, H, hd, scale)
    finally:
        ops.WIDE_BWD_FLASH = prev
    for x, y in zip(a, cmp):
        assert O.rel_l2(to_np(x), to_np(y)) < 6e-3        # (two 16-bit roundings of the outputs + where the scale enters dS)
    # size-independent properties at the path's size: a second run repeats every bit (no atomics, fixed schedules), and the
    # gradients are linear in dO - exactly so for a power of two (P does not depend on dO; dP, delta, dS and the sums double)
    a2 = ops.attention_bwd_hd(q, k, v, o, do, lse, H, hd, scale)
    d2 = ops.attention_bwd_hd(q, k, v, o, do * 2, lse, H, hd, scale)
    for x, y, z in zip(a, a2, d2):
        assert torch.equal(x, y)
        assert torch.equal(x * 2, z)
    with pytest.raises(capi.MavlmError):
        ops.attention_bwd_hd(q, k, v, o, do[:50], lse, H, hd, scale)
    with pytest.raises(capi.MavlmError):
        ops.attention_bwd_hd(q.cpu(), k, v, o, do, lse, H, hd, scale)
