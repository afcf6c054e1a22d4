"""Diagnostic: plain grid vs stream-K schedule of the attention forward at growing key counts (evolution over n cached
memories, R = 12544 queries, H = 8), same process, interleaved."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
lib = capi.lib()
R, H, D = 12544, 8, 1024
q = torch.randn(R, D, device="cuda").bfloat16()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n in (0.5, 1, 2, 5, 10):
    S = int(n * R)
    kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
    res = {}
    for rnd in range(2):
        for name, mt, wv in (("plain", 1 << 30, 0), ("sk4", 64, 4), ("sk8", 64, 8)):
            lib.mavlm_set_attention_streamk_min_tiles(mt)
            lib.mavlm_set_attention_streamk_waves(wv)
            us = t(lambda: ops.attention(q, kv[:, :D], kv[:, D:], H))
            res.setdefault(name, []).append(us)
    fl = 4.0 * R * S * D
    print(f"S={S:7d}: " + "  ".join(f"{k} {min(v):8.1f} us {fl / min(v) / 1e6:6.1f} TF" for k, v in res.items()), flush=True)
lib.mavlm_set_attention_streamk_min_tiles(64)
lib.mavlm_set_attention_streamk_waves(0)
