"""-m gpu: operator-level parity of the HIP kernels (through the C ABI) against the CPU oracle.

Gates: exact for integer-valued data (fp32 accumulation of small integers is exact);
rel-L2 <= 1e-3 against the oracle in operand-rounding-emulation mode for random data."""
import ctypes
import math

import numpy as np
import pytest
import torch

import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi
from memory_augmented_vlm_amd import _ops as ops
from oracle import memory_path as O
from gpu_util import to_dev, to_np, f32_dev

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _int_mat(shape, seed, lo=-4, hi=4):
    return np.floor(O.hash_uniform(shape, seed, lo, hi + 0.999)).astype(np.float32)


@pytest.fixture(params=[(128, 0), (256, 256), (256, 224), (257, 256), (257, 224)], ids=lambda p: f"tile{p[0]}-rows{p[1]}")
def gemm_tile(request):
    """Run the GEMM tests once per kernel: 128^2 4-wave, 256-column 8-wave LDS-DMA pipeline with 256- and 224-row
    workgroup tiles, and its persistent form (257; shapes / epilogues it does not take fall through to the
    non-persistent kernels)."""
    tile, rows = request.param
    capi.check(capi.lib().mavlm_set_gemm_tile(tile), "set tile")
    capi.check(capi.lib().mavlm_set_gemm_rows(rows), "set rows")
    yield tile
    capi.lib().mavlm_set_gemm_tile(0)
    capi.lib().mavlm_set_gemm_rows(0)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(16, 128, 64), (200, 256, 192), (129, 128, 1024), (1, 384, 64), (513, 512, 64),
                                   (300, 256, 128), (256, 768, 320), (4100, 4096, 192)])
def test_linear_integer_exact(mode, M, N, K, gemm_tile):
    """A = I-like / asymmetric integer data: catches any row/col swap in the fragment or C maps exactly."""
    A = _int_mat((M, K), 1)
    W = _int_mat((N, K), 2)
    W[:, 0] += np.arange(N) % 5          # asymmetric
    b = _int_mat((N,), 3).astype(np.float32)
    ref = A @ W.T + b
    res = _int_mat((M, N), 4)
    out = ops.linear(to_dev(A, mode), to_dev(W, mode), f32_dev(b), capi.EPI_RES_F32, residual=to_dev(res, mode))
    np.testing.assert_array_equal(to_np(out), ref + res)
    out_relu = ops.linear(to_dev(A, mode), to_dev(W, mode), f32_dev(b), capi.EPI_RELU)
    r = O.rounder(mode)
    np.testing.assert_array_equal(to_np(out_relu), r(np.maximum(ref, 0)))
    out_f32 = ops.linear(to_dev(A, mode), to_dev(W, mode), f32_dev(b), capi.EPI_F32)       # bias -> fp32 (no residual)
    np.testing.assert_array_equal(to_np(out_f32), ref)
    out_b = ops.linear(to_dev(A, mode), to_dev(W, mode), f32_dev(b), capi.EPI_BIAS)
    np.testing.assert_array_equal(to_np(out_b), r(ref))


def test_gemm_tile_height_is_a_pure_speed_choice():
    """The 224-row and 256-row workgroup tiles give bit-identical results (same K order per output element), on random
    data, for ragged M (not a multiple of either height), in both the plain and the persistent kernel."""
    lib = capi.lib()
    M, N, K = 12544 + 37, 1024, 320
    a = to_dev(O.bf16_round(O.hash_normal_like((M, K), 71)))
    w = to_dev(O.bf16_round(O.hash_uniform((N, K), 72, -0.1, 0.1)))
    b = f32_dev(O.hash_uniform((N,), 73, -0.1, 0.1))
    outs = []
    try:
        for tile in (256, 257):
            for rows in (256, 224):
                lib.mavlm_set_gemm_tile(tile)
                lib.mavlm_set_gemm_rows(rows)
                outs.append(ops.linear(a, w, b, capi.EPI_GELU).clone())
                outs.append(ops.linear(a, w, b, capi.EPI_F32).clone().view(torch.int32))
    finally:
        lib.mavlm_set_gemm_tile(0)
        lib.mavlm_set_gemm_rows(0)
    for o in outs[2::2]:
        assert torch.equal(o, outs[0])
    for o in outs[3::2]:
        assert torch.equal(o, outs[1])


def test_gemm_128x256_two_workgroups_per_cu_kernel():
    """gemm128.hip (round 4: 4 waves on a 128x256 tile, two independent workgroups per CU, a 3-slot ring per operand recycled
    at sub-piece granularity) - selected through the tuning hook mavlm_set_gemm_tile(129).  Exact against numpy on integer data
    (ragged M, 1 / 2 / 3 / many K-tiles: the prologue and the counted waits of the last K-tiles), bit-identical to the 256-row
    kernels on random data for every epilogue, both dtypes, repeated launches (LDS-DMA hazards), and the fused dense + residual
    + LayerNorm epilogue bit-identical to the 256-row fused kernel (same statistics, same merge order)."""
    lib = capi.lib()
    try:
        for (M, N, K) in [(100, 256, 64), (129, 256, 128), (777, 512, 192), (1568, 1024, 1024), (3000, 256, 4096)]:
            A = _int_mat((M, K), 31 + K)
            W = _int_mat((N, K), 32 + K)
            b = _int_mat((N,), 33).astype(np.float32)
            ref = A @ W.T + b
            lib.mavlm_set_gemm_tile(129)
            for _ in range(3):
                np.testing.assert_array_equal(to_np(ops.linear(to_dev(A), to_dev(W), f32_dev(b), capi.EPI_F32)), ref)
        for mode in ("bf16", "fp16"):
            M, N, K = 12544 + 37, 1024, 320
            a = to_dev(O.bf16_round(O.hash_normal_like((M, K), 71)), mode)
            w = to_dev(O.bf16_round(O.hash_uniform((N, K), 72, -0.1, 0.1)), mode)
            b = f32_dev(O.hash_uniform((N,), 73, -0.1, 0.1))
            for epi in (capi.EPI_BIAS, capi.EPI_RELU, capi.EPI_GELU, capi.EPI_F32):
                lib.mavlm_set_gemm_tile(256)
                ref = ops.linear(a, w, b, epi).clone()
                lib.mavlm_set_gemm_tile(129)
                for _ in range(3):
                    assert torch.equal(ops.linear(a, w, b, epi), ref), (mode, epi)
        # the Residual block in one kernel, on the 128-row tiles (row blocks of 128 rows, N / 256 partners)
        for (M, N, K) in [(25088, 1024, 256), (12544 + 5, 1024, 1024), (24576, 512, 128)]:
            a = to_dev(O.bf16_round(O.hash_normal_like((M, K), 81)))
            w = to_dev(O.bf16_round(O.hash_uniform((N, K), 82, -0.1, 0.1)))
            b = f32_dev(O.hash_uniform((N,), 83, -0.1, 0.1))
            res = to_dev(O.bf16_round(O.hash_normal_like((M, N), 84)))
            g = f32_dev(O.hash_uniform((N,), 85, 0.5, 1.5))
            be = f32_dev(O.hash_uniform((N,), 86, -0.5, 0.5))
            lib.mavlm_set_gemm_tile(0)
            ref = ops.linear_residual_layernorm(a, w, b, res, g, be, 1e-12)[0].clone()
            lib.mavlm_set_gemm_tile(129)
            for _ in range(3):
                assert torch.equal(ops.linear_residual_layernorm(a, w, b, res, g, be, 1e-12)[0], ref), (M, N, K)
    finally:
        lib.mavlm_set_gemm_tile(0)


def test_gemm_splitk_on_256_row_tiles():
    """Round 4: a long contraction (K >= 8192) over a grid of 64-128 tiles of 256 rows - the 4D -> D projection of a single
    video at the OneVision-7B width - is split into K ranges on the 256-row kernel (blockIdx.y, fp32 planes) and reduced
    once.  Exact on integer data for every epilogue the reduce applies, ragged M, and the plan is what the workspace query
    reports."""
    lib = capi.lib()
    M, N, K = 1568 + 5, 2560, 8192
    assert lib.mavlm_linear_ws_floats(M, N, K, capi.EPI_F32, N) == 3 * M * N        # 7 x 10 = 70 tiles -> 3 K ranges
    assert lib.mavlm_linear_ws_floats(M, N, 4096, capi.EPI_F32, N) == 0
    A = _int_mat((M, K), 41, -2, 2)
    W = _int_mat((N, K), 42, -2, 2)
    b = _int_mat((N,), 43).astype(np.float32)
    ref = A @ W.T + b
    a_, w_, b_ = to_dev(A), to_dev(W), f32_dev(b)
    for _ in range(2):
        np.testing.assert_array_equal(to_np(ops.linear(a_, w_, b_, capi.EPI_F32)), ref)
    np.testing.assert_array_equal(to_np(ops.linear(a_, w_, b_, capi.EPI_RELU)), O.bf16_round(np.maximum(ref, 0)))
    np.testing.assert_array_equal(to_np(ops.linear(a_, w_, b_, capi.EPI_BIAS)), O.bf16_round(ref))
    # the 7B shape itself, random data, against the unsplit kernel (fp32 partial planes: another summation order)
    M, N, K = 1568, 3584, 14336
    a = to_dev(O.bf16_round(O.hash_normal_like((M, K), 44)))
    w = to_dev(O.bf16_round(O.hash_uniform((N, K), 45, -0.01, 0.01)))
    bb = f32_dev(O.hash_uniform((N,), 46, -0.1, 0.1))
    got = ops.linear(a, w, bb, capi.EPI_F32).clone()
    try:
        lib.mavlm_set_gemm_tile(256)          # (a forced tile disables the split plan)
        want = ops.linear(a, w, bb, capi.EPI_F32).clone()
    finally:
        lib.mavlm_set_gemm_tile(0)
    assert O.rel_l2(to_np(got), to_np(want)) < 1e-5


def test_gemm_256_tile_kernels_with_l2_resident_weights():
    """The 256-column-tile kernels re-stage a B half-tile (LDS-DMA) in the barrier interval in which the other wave group
    retires its last reads of it; the margin is the DMA's latency (gemm256.hip, hazard table: WAR).  The shortest latency
    is a weight matrix that stays in L2 (small N*K) streamed over many K-tiles: integer data, exact against numpy, for both
    kernels and both tile heights, repeated - and bit-identical to the 128-tile kernel (which has no such interval)."""
    lib = capi.lib()
    M, N, K = 7168, 256, 4096                                   # W = 2 MiB: L2-resident; 64 K-tiles
    A = _int_mat((M, K), 21)
    W = _int_mat((N, K), 22)
    b = _int_mat((N,), 23).astype(np.float32)
    ref = A @ W.T + b
    a_, w_, b_ = to_dev(A), to_dev(W), f32_dev(b)
    ar = to_dev(O.bf16_round(O.hash_normal_like((M, K), 24)))
    try:
        lib.mavlm_set_gemm_tile(128)
        base = ops.linear(ar, w_, b_, capi.EPI_F32).clone()
        for tile in (256, 257):
            for rows in (256, 224):
                lib.mavlm_set_gemm_tile(tile)
                lib.mavlm_set_gemm_rows(rows)
                for _ in range(10):
                    np.testing.assert_array_equal(to_np(ops.linear(a_, w_, b_, capi.EPI_F32)), ref)
                    assert torch.equal(ops.linear(ar, w_, b_, capi.EPI_F32), base)
    finally:
        lib.mavlm_set_gemm_tile(0)
        lib.mavlm_set_gemm_rows(0)


@pytest.mark.parametrize("M,N,K,epi", [(1568, 1024, 1024, "bias"), (588, 4096, 1024, "bias"), (777, 512, 4096, "gelu"),
                                       (300, 4096, 1024, "relu")])
def test_linear_random_vs_oracle(M, N, K, epi, gemm_tile):
    r = O.bf16_round
    A = r(O.hash_normal_like((M, K), 11))
    W = r(O.hash_uniform((N, K), 12, -1 / math.sqrt(K), 1 / math.sqrt(K)))
    b = r(O.hash_uniform((N,), 13, -0.1, 0.1))
    y = O.linear(A, W, b)
    code = {"bias": capi.EPI_BIAS, "relu": capi.EPI_RELU, "gelu": capi.EPI_GELU}[epi]
    if epi == "relu":
        y = np.maximum(y, 0)
    elif epi == "gelu":
        y = O.gelu_erf(y)
    got = to_np(ops.linear(to_dev(A), to_dev(W), f32_dev(b), code))
    assert O.rel_l2(got, r(y)) < TOL


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(300, 256, 2048), (1568, 1024, 4096), (129, 128, 4160), (1, 1024, 2048)])
def test_linear_split_k_small_grids(mode, M, N, K):
    """<= 128 output tiles and K >= 2048 (the 4D -> D projections at 8 memory tokens) split the contraction over
    blockIdx.y; bias and epilogue are applied once, in the reduce.  Integer data: exact for every epilogue."""
    lib = capi.lib()
    assert lib.mavlm_linear_ws_floats(M, N, K, capi.EPI_BIAS, N) > 0
    assert lib.mavlm_linear_ws_floats(12544, 1024, 4096, capi.EPI_F32, 1024) == 0      # full grids never split
    assert lib.mavlm_linear_ws_floats(M, N, K, capi.EPI_RES_F32, N) == 0
    A = _int_mat((M, K), 1, -2, 2)
    W = _int_mat((N, K), 2, -2, 2)
    b = _int_mat((N,), 3, -8, 8)
    ref = A @ W.T + b
    r = O.rounder(mode)
    a, w, bb = to_dev(A, mode), to_dev(W, mode), f32_dev(b)
    assert np.array_equal(to_np(ops.linear(a, w, bb, capi.EPI_BIAS)), r(ref))
    assert np.array_equal(to_np(ops.linear(a, w, bb, capi.EPI_RELU)), r(np.maximum(ref, 0)))
    assert np.array_equal(to_np(ops.linear(a, w, bb, capi.EPI_F32)), ref)
    assert O.rel_l2(to_np(ops.linear(a, w, bb, capi.EPI_GELU)), r(O.gelu_erf(ref))) < 1e-4
    # a too-small workspace is an error, not a silent change of path
    out = torch.empty((M, N), device="cuda", dtype=a.dtype)
    assert lib.mavlm_linear_ws(a.data_ptr(), K, w.data_ptr(), K, bb.data_ptr(), 0, 0, out.data_ptr(), N, M, N, K,
                               capi.EPI_BIAS, 0, 0, ops.dtype_code(a.dtype), ops.stream_ptr()) == capi.E_ARG
    # random data against the oracle
    Ar = r(O.hash_normal_like((M, K), 11))
    Wr = r(O.hash_uniform((N, K), 12, -1 / math.sqrt(K), 1 / math.sqrt(K)))
    got = to_np(ops.linear(to_dev(Ar, mode), to_dev(Wr, mode), bb, capi.EPI_BIAS))
    assert O.rel_l2(got, r(O.linear(Ar, Wr, b))) < TOL


@pytest.fixture(params=[2, 3])
def attn_impl(request):
    """Run the attention tests once per forward kernel (2 = register-staged, 3 = software-pipelined LDS-DMA)."""
    capi.check(capi.lib().mavlm_set_attention_impl(request.param), "set attention impl")
    yield request.param
    capi.lib().mavlm_set_attention_impl(0)


def _attn_oracle(q, k, v, H, mode="bf16"):
    """ctx (rounded to the 16-bit grid), lse2 [H,R], per-head column sums [H,S] from the oracle's emulation of the
    kernel (64-key tiles, running maximum, P rounded per tile)."""
    ctx, lse2, col, _ = O.attention_heads(q, k, v, H, mode, want_colsum=True)
    return O.rounder(mode)(ctx), lse2, col


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("R,S,H", [(32, 64, 1), (200, 300, 2), (128, 64 * 5 + 1, 1), (392, 588, 8), (1, 1, 1), (64, 128, 1),
                                   (129, 64 * 3, 2), (300, 64 * 4 + 63, 1)])
def test_attention_vs_oracle(mode, R, S, H, attn_impl):
    r = O.rounder(mode)
    q = r(O.hash_normal_like((R, H * 128), 21))
    k = r(O.hash_normal_like((S, H * 128), 22))
    v = r(O.hash_normal_like((S, H * 128), 23))
    ctx, lse2, col = _attn_oracle(q, k, v, H, mode)
    got, lse = ops.attention(to_dev(q, mode), to_dev(k, mode), to_dev(v, mode), H, want_lse=True)
    assert O.rel_l2(to_np(got), ctx) < TOL
    np.testing.assert_allclose(to_np(lse), lse2, rtol=0, atol=2e-3)
    part = ops.attention_colsum(to_dev(q, mode), to_dev(k, mode), lse, H)
    assert O.rel_l2(to_np(part), col) < TOL
    # size-independent property: every softmax row sums to one -> all column sums add up to H*R
    assert abs(float(part.sum()) - H * R) < 1e-3 * H * R


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("R,S,H", [(300, 2000, 2), (1568, 6272, 8), (100, 64 * 16, 1), (196, 64 * 40 + 17, 4)])
def test_attention_split_kv_small_grids(mode, R, S, H):
    """Grids of fewer than 320 workgroups split the keys over blockIdx.y (normalised fp32 partials + merge kernel):
    same gates as the unsplit kernel against the oracle; the split result stays within 16-bit rounding of the
    unsplit one; the log-sum-exp it hands to the column-sum pass and to the backward is the merged one."""
    lib = capi.lib()
    assert lib.mavlm_attention_ws_floats(R, S, H) > 0 and lib.mavlm_attention_ws_floats(8192, 6272, 8) == 0
    r = O.rounder(mode)
    q = r(O.hash_normal_like((R, H * 128), 21))
    k = r(O.hash_normal_like((S, H * 128), 22))
    v = r(O.hash_normal_like((S, H * 128), 23))
    ctx, lse2, col = _attn_oracle(q, k, v, H, mode)
    dq, dk, dv = to_dev(q, mode), to_dev(k, mode), to_dev(v, mode)
    got, lse = ops.attention(dq, dk, dv, H, want_lse=True)
    assert O.rel_l2(to_np(got), ctx) < TOL
    np.testing.assert_allclose(to_np(lse), lse2, rtol=0, atol=2e-3)
    part = ops.attention_colsum(dq, dk, lse, H)
    assert O.rel_l2(to_np(part), col) < TOL and abs(float(part.sum()) - H * R) < 1e-3 * H * R
    # the unsplit kernel through the plain entry point (never splits)
    plain = torch.empty_like(got)
    lse_p = torch.empty_like(lse)
    capi.check(lib.mavlm_attention(dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0),
                                   plain.data_ptr(), plain.stride(0), lse_p.data_ptr(), R, S, H, 1.0 / math.sqrt(128.0),
                                   ops.dtype_code(dq.dtype), ops.stream_ptr()), "mavlm_attention")
    assert O.rel_l2(to_np(got), to_np(plain)) < (3e-3 if mode == "bf16" else 5e-4)
    np.testing.assert_allclose(to_np(lse), to_np(lse_p), rtol=0, atol=1e-4)
    # a too-small workspace is an error, not a silent change of path
    assert lib.mavlm_attention_ws(dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0),
                                  plain.data_ptr(), plain.stride(0), 0, R, S, H, 1.0 / math.sqrt(128.0), 0, 0,
                                  ops.dtype_code(dq.dtype), ops.stream_ptr()) == capi.E_ARG


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("R,F,P,H", [(1568, 3, 196, 8), (300, 5, 196, 2), (129, 64, 64, 1), (8330, 2, 100, 8), (4100, 7, 68, 3)])
def test_attention_frames_vs_oracle(mode, R, F, P, H):
    """Forward + frame scores in one pass (what the fused step runs for the last formation layer): the context and the
    log-sum-exp are BIT-identical to the forward without scores on the same schedule (the frame bookkeeping does not touch
    them; the small grids that `ops.attention` would split over the keys run the plain grid here), the scores match
    the oracle's column sums averaged per frame; frame boundaries inside a 64-key tile (P = 196, 100, 68) and at tile ends
    (P = 64, the smallest frame: one boundary per tile, 64 frames), ragged last tiles and query blocks.  (The stream-K
    schedule: tests/test_gpu_batch.py.)"""
    S = F * P
    r = O.rounder(mode)
    q = r(O.hash_normal_like((R, H * 128), 51)) * 2.0
    k = r(O.hash_normal_like((S, H * 128), 52))
    v = r(O.hash_normal_like((S, H * 128), 53))
    q[3] *= 8.0                                       # rows whose maximum jumps: the deferred rescale scales the frame mass too
    dq, dk, dv = to_dev(q, mode), to_dev(k, mode), to_dev(v, mode)
    assert capi.lib().mavlm_attention_frames_ws_floats(R, S, H, P) > 0 and capi.lib().mavlm_attention_frames_ws_floats(R, 32 * 60, H, 60) == 0
    got, lse, scores = ops.attention_frames(dq, dk, dv, H, P, want_lse=True)
    info = (capi.i32 * 4)()
    capi.check(capi.lib().mavlm_attention_plan(R, S, H, info), "plan")
    plain, lse_p = ops.attention(dq, dk, dv, H, want_lse=True, plain=info[3] > 1)
    assert torch.equal(got, plain) and torch.equal(lse, lse_p)
    _, _, col = _attn_oracle(q, k, v, H, mode)        # (plan-independent up to rounding: column sums of the probabilities)
    ref = col.astype(np.float64).sum(0).reshape(F, P).mean(1)
    assert O.rel_l2(to_np(scores), ref) < TOL and abs(float(scores.sum()) * P - H * R) < 1e-3 * H * R
    again = ops.attention_frames(dq, dk, dv, H, P)[2]
    assert torch.equal(again, scores)


@pytest.mark.parametrize("wgs", [0, 64, 200, 1024])
@pytest.mark.parametrize("mode,R,S,H", [("bf16", 1568, 1000, 3), ("fp16", 4100, 64 * 9 + 5, 8), ("bf16", 130, 6272, 8)])
def test_colsum_balanced_schedule(mode, R, S, H, wgs, request):
    """The column-sum pass runs G workgroups over the flattened (128-key block x head, 64-query tile) space: a workgroup's
    range may start and end in the middle of a unit and span several units; the pieces of a unit land in separate planes
    that are added in a fixed order.  Same gate against the oracle for every G (the hook changes the summation order only),
    ragged tiles / key blocks included, H not a power of two; bit-reproducible (no atomics)."""
    lib = capi.lib()
    capi.check(lib.mavlm_set_attention_colsum_wgs(wgs), "colsum wgs")
    request.addfinalizer(lambda: lib.mavlm_set_attention_colsum_wgs(0))
    assert lib.mavlm_set_attention_colsum_wgs(63) == capi.E_ARG and lib.mavlm_set_attention_colsum_wgs(wgs) == 0
    assert lib.mavlm_attention_colsum_floats(R, S, H) >= H * S
    r = O.rounder(mode)
    q = r(O.hash_normal_like((R, H * 128), 41))
    k = r(O.hash_normal_like((S, H * 128), 42))
    v = r(O.hash_normal_like((S, H * 128), 43))
    _, lse2, col = _attn_oracle(q, k, v, H, mode)
    dq, dk, dv = to_dev(q, mode), to_dev(k, mode), to_dev(v, mode)
    _, lse = ops.attention(dq, dk, dv, H, want_lse=True)
    part = ops.attention_colsum(dq, dk, lse, H)
    assert part.shape == (H, S) and O.rel_l2(to_np(part), col) < TOL and abs(float(part.sum()) - H * R) < 1e-3 * H * R
    for _ in range(3):
        assert torch.equal(ops.attention_colsum(dq, dk, lse, H), part)


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("mode,R,S,H", [("bf16", 8320, 256, 8), ("fp16", 8330, 200, 8), ("bf16", 1100, 64 * 9 + 5, 64)])
def test_attention_stream_k_more_units_than_slots(mode, R, S, H, waves, monkeypatch, request):
    """More units (query blocks x heads) than resident workgroups (the bench shape: 784 four-wave units on 512 slots): the
    levelled stream-K schedule - every workgroup takes floor(U/G) whole units, the remainder is served in binary levels whose
    units are cut into 2^k key ranges (normalised fp32 partials, merged by attn_combine_sk_kernel); 4-wave (G = 512, 128
    queries per unit) and 8-wave (G = 256, 256 queries) workgroups.  The oracle mirrors the cuts (streamk_split_tiles): same
    gates as the plain kernel; ragged query blocks / key tiles included; the plain entry point agrees within rounding."""
    lib = capi.lib()
    assert lib.mavlm_attention_ws_floats(12544, 63 * 64, 8) == 0 and lib.mavlm_attention_ws_floats(12544, 6272, 8) > 0
    capi.check(lib.mavlm_set_attention_streamk_min_tiles(1), "min tiles")     # (the default, 64 tiles, needs S > 4032)
    capi.check(lib.mavlm_set_attention_streamk_waves(waves), "waves")           # 4: 512 workgroups x 128 queries, 8: 256 x 256
    monkeypatch.setattr(O, "STREAMK_MIN_TILES", 1)
    monkeypatch.setattr(O, "STREAMK_WAVES", waves)
    request.addfinalizer(lambda: (lib.mavlm_set_attention_streamk_min_tiles(64), lib.mavlm_set_attention_streamk_waves(0)))
    assert lib.mavlm_attention_ws_floats(R, S, H) % (512 * (128 * 128 + 128)) == 0 and lib.mavlm_attention_ws_floats(R, S, H) > 0
    assert O.streamk_wgs(R, S, H) == 2048 // waves and len(O.streamk_split_tiles(R, S, H)[1]) >= 8
    r = O.rounder(mode)
    q = r(O.hash_normal_like((R, H * 128), 31))
    k = r(O.hash_normal_like((S, H * 128), 32))
    v = r(O.hash_normal_like((S, H * 128), 33))
    q[5] *= 8.0                                       # a few rows whose maximum jumps (deferred-rescale branch);
    k[S // 2 + 3] *= 4.0                              # powers of two: the data stay on the 16-bit grid
    ctx, lse2, col = _attn_oracle(q, k, v, H, mode)
    dq, dk, dv = to_dev(q, mode), to_dev(k, mode), to_dev(v, mode)
    got, lse = ops.attention(dq, dk, dv, H, want_lse=True)
    assert O.rel_l2(to_np(got), ctx) < TOL
    np.testing.assert_allclose(to_np(lse), lse2, rtol=0, atol=2e-3)
    again, _ = ops.attention(dq, dk, dv, H, want_lse=True)
    assert torch.equal(again, got)                    # static schedule, no atomics: deterministic
    part = ops.attention_colsum(dq, dk, lse, H)
    assert O.rel_l2(to_np(part), col) < TOL and abs(float(part.sum()) - H * R) < 1e-3 * H * R
    plain = torch.empty_like(got)
    lse_p = torch.empty_like(lse)
    capi.check(lib.mavlm_attention(dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0),
                                   plain.data_ptr(), plain.stride(0), lse_p.data_ptr(), R, S, H, 1.0 / math.sqrt(128.0),
                                   ops.dtype_code(dq.dtype), ops.stream_ptr()), "mavlm_attention")
    assert O.rel_l2(to_np(got), to_np(plain)) < (3e-3 if mode == "bf16" else 5e-4)
    np.testing.assert_allclose(to_np(lse), to_np(lse_p), rtol=0, atol=1e-4)


def test_attention_strided_kv_and_identity_v(attn_impl):
    """K/V as column slices of a wider [S, 4D] buffer (how the step lays them out) and V = one-hot columns:
    ctx then equals the probabilities themselves - catches any key/column permutation error in the P.V MFMA."""
    R, S, H = 64, 128, 2
    r = O.bf16_round
    q = r(O.hash_normal_like((R, H * 128), 31))
    kvbuf = np.zeros((S, 4 * H * 128), np.float32)
    k = r(O.hash_normal_like((S, H * 128), 32))
    v = np.zeros((S, H * 128), np.float32)
    for h in range(H):
        v[np.arange(S), h * 128 + (np.arange(S) * 7 + 3 * h) % 128] = 1.0      # asymmetric one-hot
    kvbuf[:, :H * 128] = k
    kvbuf[:, 2 * H * 128:3 * H * 128] = v
    t = to_dev(kvbuf)
    got, _ = ops.attention(to_dev(q), t[:, :H * 128], t[:, 2 * H * 128:3 * H * 128], H)
    ctx, _, _ = _attn_oracle(q, k, v, H)
    assert O.rel_l2(to_np(got), ctx) < TOL


def test_attention_forced_rescale(attn_impl):
    """Rule 26: force the online-softmax rescale branch - one key in a LATE tile dominates one query row."""
    R, S, H = 64, 64 * 6, 1
    r = O.bf16_round
    q = r(O.hash_normal_like((R, 128), 41))
    k = r(O.hash_normal_like((S, 128), 42))
    v = r(O.hash_normal_like((S, 128), 43))
    k[64 * 4 + 17] = r(q[5] * 4.0)      # spike in tile 4 for query 5
    k[64 * 5 + 63] = r(q[40] * 6.0)     # spike in the last tile for query 40
    ctx, lse2, _ = _attn_oracle(q, k, v, H)
    got, lse = ops.attention(to_dev(q), to_dev(k), to_dev(v), H, want_lse=True)
    assert O.rel_l2(to_np(got), ctx) < TOL
    assert np.abs(to_np(got)[[5, 40]] - ctx[[5, 40]]).max() < 2e-2
    np.testing.assert_allclose(to_np(lse), lse2, rtol=0, atol=5e-3)


@pytest.mark.parametrize("D", [128, 256, 896, 1024, 3584])
def test_layernorm_vs_oracle(D):
    rows = 37
    x = O.hash_normal_like((rows, D), 51) * 3.0 + 0.5
    g = 1.0 + O.hash_uniform((D,), 52, -0.1, 0.1)
    b = O.hash_uniform((D,), 53, -0.1, 0.1)
    ref = O.bf16_round(O.layernorm(x, g, b, 1e-12))
    got = to_np(ops.layernorm(f32_dev(x), f32_dev(g), f32_dev(b), 1e-12, torch.bfloat16))
    assert O.rel_l2(got, ref) < 2e-4
    # fused residual add (fp32) of the Residual block
    res = O.bf16_round(O.hash_normal_like((rows, D), 54))
    ref_r = O.bf16_round(O.layernorm(x + res, g, b, 1e-12))
    got_r = to_np(ops.layernorm(f32_dev(x), f32_dev(g), f32_dev(b), 1e-12, torch.bfloat16, residual=to_dev(res)))
    assert O.rel_l2(got_r, ref_r) < 2e-4
    # constant rows: var = 0 -> rsqrt(0 + 1e-12) finite, output = beta
    xc = np.full((4, D), 2.5, np.float32)
    gc = to_np(ops.layernorm(f32_dev(xc), f32_dev(g), f32_dev(b), 1e-12, torch.bfloat16))
    np.testing.assert_array_equal(gc, O.bf16_round(np.broadcast_to(b, (4, D))))


def test_row_add_exact():
    T, P, D = 9, 196, 256
    x = O.bf16_round(O.hash_normal_like((T, P, D), 61))
    table = O.bf16_round(O.hash_normal_like((50, D), 62))
    idx = np.array([0, 49, 7, 7, 13, 2, 48, 1, 30])
    ref = O.bf16_round(x + table[idx][:, None, :])
    got = ops.row_add(to_dev(x), to_dev(table), idx=torch.from_numpy(idx).cuda())
    np.testing.assert_array_equal(to_np(got), ref)
    src = np.array([8, 0, 3])
    got2 = ops.row_add(to_dev(x), to_dev(table[5:6]), src=torch.from_numpy(src).cuda())
    np.testing.assert_array_equal(to_np(got2), O.bf16_round(x[src] + table[5][None, None, :]))
    assert ops.row_add(to_dev(x[:0]), to_dev(table), idx=torch.zeros(0, dtype=torch.int64).cuda()).shape[0] == 0


def test_bad_arguments_fail_loudly():
    a = to_dev(np.zeros((4, 64), np.float32))
    w = to_dev(np.zeros((100, 64), np.float32))     # N not a multiple of 128
    with pytest.raises(capi.MavlmError):
        ops.linear(a, w, f32_dev(np.zeros(100, np.float32)))
    with pytest.raises(capi.MavlmError):
        ops.linear(a.cpu(), w.cpu(), torch.zeros(100))


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_pool_bilinear_vs_oracle_and_reference_golden(mode):
    """get_2dPool bilinear 27x27 -> 14x14 (llava_arch.py:277-297).  The oracle's bilinear_pool is pinned to the
    reference's own output (tests/golden/g6_glue.npz: pooled_2); here the HIP kernel is checked against it."""
    from conftest import load_golden
    r = O.rounder(mode)
    side, D, F = 27, 256, 5
    x = r(O.hash_normal_like((F, side * side, D), 71))
    ref = r(O.bilinear_pool(x, side))
    got = to_np(ops.pool_bilinear(to_dev(x, mode), side))
    ulp = 2.0 ** (-7 if mode == "bf16" else -10)      # spacing of the grid in [1, 2): 7 / 10 stored mantissa bits
    assert np.abs(got - ref).max() <= ulp * np.abs(ref).max() * 1.01          # at most one 16-bit ulp
    assert O.rel_l2(got, ref) < 3e-4
    # fused PE add keeps the two reference roundings
    table = r(O.pe_table(50, D))
    idx = np.array([0, 49, 7, 7, 13])
    ref2 = r(ref + table[idx][:, None, :])
    got2 = to_np(ops.pool_bilinear(to_dev(x, mode), side, pe_table=to_dev(table, mode), idx=torch.from_numpy(idx).cuda()))
    assert O.rel_l2(got2, ref2) < 3e-4
    if mode == "bf16":   # the reference's own pooled output (fp32 run on bf16-grid inputs), D = 32
        z, m = load_golden("g6_glue.npz")
        feats = O.bf16_round(O.hash_normal_like((8, side * side, m["D"]), int(z["pool_in_seed"])))[:2]
        got3 = to_np(ops.pool_bilinear(to_dev(feats), side))
        assert O.rel_l2(got3, z["pooled_2"]) < 3e-3                            # bf16 output rounding vs fp32 reference


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("hd,R,S,H", [(128, 200, 300, 2), (448, 16, 32, 1), (448, 200, 300, 2), (448, 129, 97, 1),
                                      (448, 392, 588, 8), (128, 1, 1, 1), (448, 1, 1, 1), (448, 196, 2048, 2),
                                      (128, 100, 1100, 1), (448, 1568, 6272, 8)])
@pytest.mark.parametrize("groups", [2, 1])
def test_attention_wide_heads_vs_oracle(mode, hd, R, S, H, groups):
    """attention_hd.hip: head_dim 448 (LLaVA-OneVision-7B) with the pipelined 32-query waves (groups = 2, the default) and
    with the 16-query waves of rounds 1-2 (groups = 1), and 128 (cross-check of the same machinery); oracle emulation with
    32-key tiles and the deferred-rescale decision per 16 / 32 query rows.  The last three
    shapes are small grids with many key tiles: the kernel splits the keys (partials + merge), the oracle mirrors the
    plan."""
    if hd != 448 and groups == 1:
        pytest.skip("one kernel form below 448")
    capi.lib().mavlm_set_attention_wide_groups(groups)
    try:
        _wide_heads_vs_oracle(mode, hd, R, S, H, 32 if (groups == 2 and hd == 448) else 16)
    finally:
        capi.lib().mavlm_set_attention_wide_groups(0)


def _wide_heads_vs_oracle(mode, hd, R, S, H, wave_rows):
    r = O.rounder(mode)
    W = H * hd
    q = r(O.hash_normal_like((R, W), 81))
    k = r(O.hash_normal_like((S, W), 82))
    v = r(O.hash_normal_like((S, W), 83))
    ctx, lse2, col, _ = O.attention_heads(q, k, v, H, mode, want_colsum=True, kv_tile=32, wave_rows=wave_rows)
    got, lse = ops.attention(to_dev(q, mode), to_dev(k, mode), to_dev(v, mode), H, want_lse=True, head_dim=hd,
                             wide_kernel=True)
    assert O.rel_l2(to_np(got), r(ctx)) < TOL
    np.testing.assert_allclose(to_np(lse), lse2, rtol=0, atol=3e-3)
    part = ops.attention_colsum(to_dev(q, mode), to_dev(k, mode), lse, H, head_dim=hd, wide_kernel=True)
    assert O.rel_l2(to_np(part), col) < TOL
    assert abs(float(part.sum()) - H * R) < 1e-3 * H * R


@pytest.mark.parametrize("mode,R,S,H", [("bf16", 4224, 1024 + 17, 8), ("fp16", 6100, 800, 8), ("bf16", 1500, 700, 23)])
def test_attention_wide_heads_stream_k(mode, R, S, H, monkeypatch, request):
    """head_dim 448 with more 128-query units than the 256 workgroups of one-per-CU (the OneVision-7B row batch: 416): the
    levelled stream-K plan of attn_fwd_hd2_kernel - whole units first, the remainder in binary levels whose units are cut into
    2^k key ranges (normalised fp32 partials, merged by attn_combine_hd_sk_kernel).  The oracle mirrors the cuts
    (streamk_split_tiles_wide); ragged query blocks / key tiles, a 32-way level (264 units), a 2-way level (384 units) and a
    mixed one (276 units) included; deterministic; agrees with the plain grid within rounding."""
    lib = capi.lib()
    info = (ctypes.c_int32 * 4)()
    capi.check(lib.mavlm_attention_hd_plan_info(12544, 6272, 8, 448, info), "plan info")
    assert list(info) == [256, 3, 1, 1]                                   # 784 units: 3 whole rounds + 16 units cut 16-way
    capi.check(lib.mavlm_attention_hd_plan_info(1568, 6272, 32, 448, info), "plan info")
    assert list(info)[:3] == [256, 1, 2]                                  # the OneVision-7B batch of four: 416 units
    capi.check(lib.mavlm_attention_hd_plan_info(1568, 6272, 8, 448, info), "plan info")
    assert info[0] == 0 and info[3] > 1                                   # a single video: small grid, keys split
    capi.check(lib.mavlm_set_attention_streamk_min_tiles(8), "min tiles")  # (the default needs 4 096 keys)
    monkeypatch.setattr(O, "STREAMK_MIN_TILES", 8)
    request.addfinalizer(lambda: lib.mavlm_set_attention_streamk_min_tiles(64))
    capi.check(lib.mavlm_attention_hd_plan_info(R, S, H, 448, info), "plan info")
    G, full, levels = O.streamk_plan_wide(R, S, H)
    assert info[0] == 256 == G and info[1] == full and info[2] == len(levels) and len(O.streamk_split_tiles_wide(R, S, H)) >= 8
    assert lib.mavlm_attention_hd_ws_floats(R, S, H, 448) == 256 * len(levels) * (128 * 448 + 128)
    r = O.rounder(mode)
    W = H * 448
    q = r(O.hash_normal_like((R, W), 131))
    k = r(O.hash_normal_like((S, W), 132))
    v = r(O.hash_normal_like((S, W), 133))
    q[5] *= 8.0                                       # a few rows whose maximum jumps (deferred-rescale branch);
    k[S // 2 + 3] *= 4.0                              # powers of two: the data stay on the 16-bit grid
    ctx, lse2, _, _ = O.attention_heads(q, k, v, H, mode, kv_tile=32, wave_rows=32)
    dq, dk, dv = to_dev(q, mode), to_dev(k, mode), to_dev(v, mode)
    got, lse = ops.attention(dq, dk, dv, H, want_lse=True, head_dim=448, wide_kernel=True)
    assert O.rel_l2(to_np(got), r(ctx)) < TOL
    np.testing.assert_allclose(to_np(lse), lse2, rtol=0, atol=3e-3)
    again, lse_again = ops.attention(dq, dk, dv, H, want_lse=True, head_dim=448, wide_kernel=True)
    assert torch.equal(again, got) and torch.equal(lse_again, lse)       # static schedule, no atomics: deterministic
    plain = torch.empty_like(got)
    lse_p = torch.empty_like(lse)
    capi.check(lib.mavlm_attention_hd(dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0),
                                      plain.data_ptr(), plain.stride(0), lse_p.data_ptr(), R, S, H, 448, 1.0 / math.sqrt(448.0),
                                      ops.dtype_code(dq.dtype), ops.stream_ptr()), "mavlm_attention_hd")
    assert O.rel_l2(to_np(got), to_np(plain)) < (3e-3 if mode == "bf16" else 5e-4)
    np.testing.assert_allclose(to_np(lse), to_np(lse_p), rtol=0, atol=1e-4)


@pytest.mark.parametrize("groups", [2, 1])
def test_attention_wide_heads_identity_v_and_rescale(groups):
    """One-hot V (ctx = probabilities: catches key/column permutation errors of the P.V operand mapping at 448) and a
    late dominating key (forces the deferred-rescale branch - in the 32-query form on an O^T that sits in the accumulator
    file)."""
    capi.lib().mavlm_set_attention_wide_groups(groups)
    try:
        _wide_identity_v(groups)
    finally:
        capi.lib().mavlm_set_attention_wide_groups(0)


def _wide_identity_v(groups):
    R, S, H, hd = 48, 160, 1, 448
    r = O.bf16_round
    q = r(O.hash_normal_like((R, hd), 91) * 0.3)
    k = r(O.hash_normal_like((S, hd), 92) * 0.3)
    v = np.zeros((S, hd), np.float32)
    v[np.arange(S), (np.arange(S) * 11 + 5) % hd] = 1.0
    k[32 * 4 + 17] = r(q[5] * 6.0)
    ctx, lse2, _, _ = O.attention_heads(q, k, v, H, "bf16", kv_tile=32, wave_rows=32 if groups == 2 else 16)
    got, lse = ops.attention(to_dev(q), to_dev(k), to_dev(v), H, want_lse=True, head_dim=hd)
    assert O.rel_l2(to_np(got), r(ctx)) < TOL
    np.testing.assert_allclose(to_np(lse), lse2, rtol=0, atol=5e-3)
