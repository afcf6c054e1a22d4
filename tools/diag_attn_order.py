"""Diagnostic (GPU box), round 4: XCD-affine unit order of the levelled stream-K attention schedule (mavlm_set_attention_unit_order
1) against the position order of rounds 1-3 (0).  The bench launch shape: 16 (video, head) pairs x 12 544 queries (784 units of
256 queries on 256 workgroups), 6 272 / 12 544 keys; plain forward and the frame-score variant; single video (H = 8).
Interleaved timing in one process; outputs of the two orders compared (different units are cut: equal within rounding).
usage: python tools/diag_attn_order.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from diag_vs_hipblaslt_util import timeit_pair

lib = capi.lib()
R = 12544
for H, S in [(16, 6272), (16, 12544), (8, 6272), (8, 12544), (8, 125440)]:
    D = H * 128
    q = torch.randn(R, D, device="cuda").bfloat16()
    kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
    k, v = kv[:, :D], kv[:, D:]
    outs = {}
    for order in (1, 0):
        lib.mavlm_set_attention_unit_order(order)
        outs[order] = ops.attention(q, k, v, H)[0].clone()
    torch.cuda.synchronize()
    d = ((outs[0].float() - outs[1].float()).norm() / outs[0].float().norm()).item()

    def new():
        lib.mavlm_set_attention_unit_order(1)
        ops.attention(q, k, v, H)

    def old():
        lib.mavlm_set_attention_unit_order(0)
        ops.attention(q, k, v, H)
    t1, t0 = timeit_pair(new, old, n=10)
    fl = 4.0 * R * S * D
    line = f"H{H:3d} S{S:7d} plain : affine {t1*1e6:8.1f} us {fl/t1/1e12:7.1f} TF | position order {t0*1e6:8.1f} us {fl/t0/1e12:7.1f} TF | {t0/t1:.3f}x | rel diff {d:.1e}"
    if S <= 12544 and S % 196 == 0 and S // 196 <= 64:
        def newf():
            lib.mavlm_set_attention_unit_order(1)
            ops.attention_frames(q, k, v, H, 196)

        def oldf():
            lib.mavlm_set_attention_unit_order(0)
            ops.attention_frames(q, k, v, H, 196)
        f1, f0 = timeit_pair(newf, oldf, n=10)
        line += f" || frames: affine {f1*1e6:8.1f} us {fl/f1/1e12:7.1f} TF | position {f0*1e6:8.1f} us | {f0/f1:.3f}x | frames/plain {f1/t1:.3f}"
    print(line, flush=True)
lib.mavlm_set_attention_unit_order(1)
