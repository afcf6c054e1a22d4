"""Diagnostic: cost of carrying the frame masses in the forward (plain grid, R = 12544, S = 6272, H = 8, P = 196)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _ops as ops
R, S, H, D = 12544, 6272, 8, 1024
q = torch.randn(R, D, device="cuda").bfloat16(); kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
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
for _ in range(2):
    a = t(lambda: ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True))
    b = t(lambda: ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True, plain=True))
    c = t(lambda: ops.attention_frames(q, kv[:, :D], kv[:, D:], H, 196, want_lse=True))
    c1 = t(lambda: ops.attention_frames(q, kv[:, :D], kv[:, D:], H, 6272, want_lse=True))     # one frame: no boundary inside
    c2 = t(lambda: ops.attention_frames(q, kv[:, :D], kv[:, D:], H, 3136, want_lse=True))
    o, lse = ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True)
    d = t(lambda: ops.attention_colsum(q, kv[:, :D], lse, H))
    print(f"stream-K + merge {a:.1f} us | plain grid {b:.1f} | plain grid + frame masses + finish {c:.1f} (1 frame {c1:.1f}, 2 frames {c2:.1f}) | column-sum pass {d:.1f}", flush=True)
