"""Driven under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: the bench attention launch (16 (video, head) pairs, 12 544 queries, 6 272 keys)
with the XCD-affine unit order (first 4 launches of each kernel) and the position order (next 4); plain and frame-score variant."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
lib = capi.lib()
R, H, S = 12544, 16, 6272
D = H * 128
q = torch.randn(R, D, device="cuda").bfloat16()
kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
for order in (1, 0):
    lib.mavlm_set_attention_unit_order(order)
    for _ in range(4):
        ops.attention(q, kv[:, :D], kv[:, D:], H)
    for _ in range(4):
        ops.attention_frames(q, kv[:, :D], kv[:, D:], H, 196)
    torch.cuda.synchronize()
lib.mavlm_set_attention_unit_order(1)
