"""Diagnostic: how much faster does a wave of the attention forward run when it has its SIMD to itself?  Plain grid, S = 6272,
H = 8: R = 4096 -> 256 four-wave workgroups (one wave per SIMD), R = 8192 -> 512 (two waves per SIMD, the normal state)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _ops as ops
S, H, D = 6272, 8, 1024
kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
def t(fn, n=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[2] * 1e3
for R in (2048, 4096, 8192, 12288, 16384):
    q = torch.randn(R, D, device="cuda").bfloat16()
    us = t(lambda: ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True, plain=True))
    print(f"R={R:6d} workgroups={R // 128 * H:5d} {us:7.1f} us  {4.0 * R * S * D / us / 1e6:7.1f} TF", flush=True)
