"""Diagnostic (GPU box): repetition screen of the kernels that got new synchronisation in round 4 - the head_dim-448 backward (LDS
exchange between the waves of a row group), the tile-entry frame scores (entry stores left in flight across the tile barrier) at both
head sizes, the event-ordered chunk projection on a side stream.  Every repetition must reproduce the first run bit for bit.
usage: python tools/diag_r4_race_screen.py [repetitions]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from memory_augmented_vlm_amd.model import llava_arch as arch
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
torch.manual_seed(0)
bad = 0
# 1. attention_bwd_hd
hd, H, R, S = 448, 8, 1568, 1568 + 32 * 196
W = H * hd
q = (torch.randn(R, W, device=dev) * 0.5).bfloat16(); k = (torch.randn(S, W, device=dev) * 0.5).bfloat16()
v = torch.randn(S, W, device=dev).bfloat16(); do = (torch.randn(R, W, device=dev) * 0.5).bfloat16()
o, lse = ops.attention(q, k, v, H, want_lse=True, head_dim=hd)
ref = ops.attention_bwd_hd(q, k, v, o, do, lse, H, hd, ops.attn_scale(hd))
for i in range(N):
    got = ops.attention_bwd_hd(q, k, v, o, do, lse, H, hd, ops.attn_scale(hd))
    if not all(torch.equal(a, b) for a, b in zip(got, ref)):
        bad += 1
print(f"attention_bwd_hd: {N} repetitions, {bad} mismatches", flush=True)
# 2./3. whole videos: M = 8 at D = 1024 (tile entries on the split-KV grid) and at D = 3584 (tile entries + side-stream projection)
for (hidden, frames, reps) in ((1024, 64, N), (3584, 128, max(4, N // 10))):
    bench.HIDDEN, bench.FRAMES = hidden, frames
    model, arch_ = bench.build_model(dev, hidden=hidden, mem_tokens=8, seed=4321)
    idx = torch.arange(frames)
    x = torch.randn((frames, bench.PATCHES, hidden), device=dev).to(torch.bfloat16)
    mem_ids = torch.tensor(arch_.MEMORY_PROMPT_IDS, device=dev); frame_ids = torch.tensor(arch_.FRAME_PROMPT_IDS, device=dev)
    rm = model.recurrent_memory_transformer
    with torch.no_grad():
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        ref, _ = arch_.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
        ref = ref.clone(); rs = [s.clone() for s in rm.frame_attn_scores[-(frames // 32):]]
        b2 = 0
        for i in range(reps):
            got, _ = arch_.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
            sc = rm.frame_attn_scores[-(frames // 32):]
            if not torch.equal(got, ref) or not all(torch.equal(a, b) for a, b in zip(sc, rs)):
                b2 += 1
            del rm.frame_attn_scores[:-8]
    print(f"video D={hidden} M=8 {frames} frames: {reps} repetitions, {b2} mismatches", flush=True)
    bad += b2
    del model
sys.exit(1 if bad else 0)
