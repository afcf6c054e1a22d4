"""Diagnostic (not a test): per-op error level and 1-ulp flip fraction, GPU vs oracle, bf16 vs fp16."""
import math, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))   # gpu_util
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from oracle import memory_path as O
from gpu_util import to_dev, to_np, f32_dev

for mode in ("bf16", "fp16"):
    r = O.rounder(mode)
    for (M, N, K) in [(1568, 1024, 1024), (1568, 1024, 4096)]:
        A = r(O.hash_normal_like((M, K), 11)); W = r(O.hash_uniform((N, K), 12, -1/math.sqrt(K), 1/math.sqrt(K)))
        b = r(O.hash_uniform((N,), 13, -0.1, 0.1))
        y32 = O.linear(A, W, b)
        y64 = (A.astype(np.float64) @ W.astype(np.float64).T + b).astype(np.float32)
        got = to_np(ops.linear(to_dev(A, mode), to_dev(W, mode), f32_dev(b), capi.EPI_BIAS))
        res = np.zeros((M, N), np.float32)
        got32 = to_np(ops.linear(to_dev(A, mode), to_dev(W, mode), f32_dev(b), capi.EPI_RES_F32, residual=to_dev(res, mode)))
        print(mode, "gemm", (M, N, K), "gpu16 vs r(cpu32): %.2e flips %.4f | r(cpu64) vs r(cpu32): %.2e flips %.4f | gpu_f32 vs cpu64: %.2e cpu32 vs cpu64: %.2e" % (
            O.rel_l2(got, r(y32)), np.mean(got != r(y32)), O.rel_l2(r(y64), r(y32)), np.mean(r(y64) != r(y32)),
            O.rel_l2(got32, y64), O.rel_l2(y32, y64)))
    R, S, H = 1568, 588, 8
    q = r(O.hash_normal_like((R, H*128), 21)); k = r(O.hash_normal_like((S, H*128), 22)); v = r(O.hash_normal_like((S, H*128), 23))
    ctx, lse2, col, _ = O.attention_heads(q, k, v, H, mode, want_colsum=True)
    got, lse = ops.attention(to_dev(q, mode), to_dev(k, mode), to_dev(v, mode), H, want_lse=True)
    print(mode, "attn", "gpu vs emu: %.2e flips %.4f ; lse maxabs %.2e" % (O.rel_l2(to_np(got), r(ctx)), np.mean(to_np(got) != r(ctx)), np.abs(to_np(lse)-lse2).max()))
