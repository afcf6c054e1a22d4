"""Diagnostic (GPU box): K ranges of the Residual block's fp32 dense output on small grids with K = 1024 (mavlm_set_gemm_short_splits:
1 = none, 2, 4), reduced inside the LayerNorm kernel: single-video latency at 8 memory tokens, D = 1024; interleaved over rounds.  usage: python tools/diag_short_splits_ab.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from memory_augmented_vlm_amd import _capi as capi  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
lib = capi.lib()
bench.HIDDEN, bench.FRAMES = 1024, 64
model, arch = bench.build_model(dev, hidden=1024, mem_tokens=8, seed=4321)          # (created under the default plan: its workspace covers all)
idx = torch.arange(64)
x = torch.randn((64, bench.PATCHES, 1024), device=dev).to(torch.bfloat16)
mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev)
frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
pool = arch.MemoryPathPool(model, 1, batch=1)


def step():
    mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
    fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
    return pool.run([(x, idx)], mp, fp, model.image_newline)


res = {sp: [] for sp in (1, 2, 4)}
with torch.no_grad():
    for sp in res:
        lib.mavlm_set_gemm_short_splits(sp)
        for _ in range(5):
            step()
    for rnd in range(6):
        for sp in res:
            lib.mavlm_set_gemm_short_splits(sp)          # (the plan is read at every launch)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40):
                step()
            torch.cuda.synchronize(); res[sp].append((time.perf_counter() - t0) / 40)
lib.mavlm_set_gemm_short_splits(2)
for sp, v in res.items():
    v.sort()
    print(f"short splits {sp}: {v[len(v) // 2] * 1e3:8.4f} ms per video (min {v[0] * 1e3:.4f})", flush=True)
