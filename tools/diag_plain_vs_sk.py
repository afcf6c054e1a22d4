import os, sys, torch
sys.path.insert(0, "/root/repo")
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
    print("stream-K (8-wave) + merge: %.1f us   plain 4-wave grid: %.1f us" % (
        t(lambda: ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True)),
        t(lambda: ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True, plain=True))))
