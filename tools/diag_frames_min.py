"""Diagnostic: the fused frame-score path against the column-sum pass on one small step (same engine, same inputs)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi
from memory_augmented_vlm_amd.model.memory_module.MemoryController import Config, TransformerProjector

M = int(os.environ.get("MEM_TOKENS", "8")); F = int(os.environ.get("FRAMES", "3"))
c = Config(); c.mm_hidden_size = 1024; c.mm_intermediate_size = 4096; c.mm_num_attention_heads = 8
c.num_memory_tokens = M; c.patch_size = 196; c.depth = 2; c.mm_dtype = torch.float32
torch.manual_seed(0)
rm = TransformerProjector(c).cuda().to(torch.bfloat16).eval()
x = (torch.randn(F, 196, 1024, device="cuda") * 0.5).bfloat16()
out = {}
for mode in (0, 1):
    capi.check(capi.lib().mavlm_set_frame_score_mode(mode), "mode")
    print("mode", mode, "...", flush=True)
    with torch.no_grad():
        rm.memory_cache = []
        cache, scores = rm(x)
        torch.cuda.synchronize()
    out[mode] = (cache[-1].float().cpu().numpy(), scores[-1].float().cpu().numpy())
    print("mode", mode, "scores", out[mode][1][:6], "sum", out[mode][1].sum(), flush=True)
print("memory identical:", np.array_equal(out[0][0], out[1][0]), " scores rel diff:",
      np.abs(out[0][1] - out[1][1]).max() / np.abs(out[0][1]).max())

# ---- inspect the scratch of the fused path (last step ran in mode 1)
if os.environ.get("FRAMES_DEBUG"):
    eng = rm._engine if hasattr(rm, "_engine") else rm.engine
    tot = int(capi.lib().mavlm_workspace_bytes(eng.c)) if hasattr(eng, "c") else None
    H, R, P = 8, M * 196, 196
    Fmax = eng.c.max_chunk_frames
    al = lambda v: (v + 255) & ~255
    fout_b = al(H * ((R + 127) // 128) * 4 * Fmax * 4)
    fscr_b = al(H * R * Fmax * 2 * 4)
    ws = eng.workspace[eng.workspace_base_offset:eng.workspace_base_offset + tot]
    fout = ws[tot - fout_b:tot].view(torch.float32)[:H * ((R + 127) // 128) * 4 * F].view(-1, F)
    fscr = ws[tot - fout_b - fscr_b:tot - fout_b].view(torch.float32)[:H * R * F * 2].view(H, R, F, 2)
    print("fout rows", fout.shape, fout[:3], "sum/P", fout.sum(0) / P)
    print("fscr[h0, q0]", fscr[0, 0], "fscr[h3, q100]", fscr[3, 100])
    v = eng.workspace_views()
    print("lse2[0,0], lse2[3,100]", v["lse2"][0, 0].item(), v["lse2"][3, 100].item())
