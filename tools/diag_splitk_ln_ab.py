"""Diagnostic (GPU box): the split-K reduction inside the LayerNorm kernel (mavlm_set_splitk_layernorm(1), default) against the
reduction pass + fp32 dense output + LayerNorm kernel (0): single-video latency at 8 memory tokens, D = 1024 and 3584; same
process, interleaved.  usage: python tools/diag_splitk_ln_ab.py"""
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
for (hidden, frames, M) in ((1024, 64, 8), (3584, 256, 8), (1024, 64, 16)):
    bench.HIDDEN, bench.FRAMES = hidden, frames
    model, arch = bench.build_model(dev, hidden=hidden, mem_tokens=M, seed=4321)
    idx = torch.arange(frames)
    x = torch.randn((frames, bench.PATCHES, hidden), device=dev).to(torch.bfloat16)
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
    pool = arch.MemoryPathPool(model, 1, batch=1)

    def step():
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        return pool.run([(x, idx)], mp, fp, model.image_newline)
    res = {1: [], 0: []}
    n = 30 if hidden == 1024 else 3
    with torch.no_grad():
        for mode in (1, 0):
            lib.mavlm_set_splitk_layernorm(mode)
            for _ in range(3):
                step()
        for rnd in range(5):
            for mode in (1, 0):
                lib.mavlm_set_splitk_layernorm(mode)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    step()
                torch.cuda.synchronize(); res[mode].append((time.perf_counter() - t0) / n)
    lib.mavlm_set_splitk_layernorm(1)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    print(f"D={hidden} frames={frames} M={M}: one kernel {med[1] * 1e3:8.3f} ms | three {med[0] * 1e3:8.3f} ms | {med[0] / med[1]:.4f}x", flush=True)
    del pool, model
