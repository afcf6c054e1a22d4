#!/usr/bin/env python3
"""Quick start on one MI355X: the memory path on random weights - inference, throughput mode, hipGraph, training.

    python examples/quickstart.py

Mirrors what llava/model/llava_arch.py:502-557 of the reference does for one video (PE add -> 32-frame chunks through
the recurrent memory transformer -> Memory-Fuser MLP -> token block), with the modules of this package standing in
for the reference's (same class names and state-dict keys; see INTEGRATION.md for the three import lines to change).
"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd  # noqa: F401,E402  (builds the HIP library on first import if needed)
from memory_augmented_vlm_amd.model import llava_arch as arch  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = types.SimpleNamespace(hidden_size=1024, num_memory_tokens=8)
    torch.manual_seed(0)
    base = type("Base", (torch.nn.Module,), {"__init__": lambda self, c: torch.nn.Module.__init__(self)})
    Model = type("Model", (arch.LlavaMetaModel, base), {})
    model = Model(cfg)
    model.embed_tokens = torch.nn.Embedding(50000, cfg.hidden_size)
    model.image_newline = torch.nn.Parameter(torch.zeros(cfg.hidden_size))
    model = model.to(dev).to(torch.bfloat16)

    T = 64                                                     # frames after the reference's sub-sampling
    frames = (torch.randn(T, 196, cfg.hidden_size, device=dev) * 0.5).bfloat16()   # pooled SigLIP tokens [T,196,D]
    idx = torch.arange(T)                                      # original frame indices (host)
    mp = model.embed_tokens(torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev))
    fp = model.embed_tokens(torch.tensor(arch.FRAME_PROMPT_IDS, device=dev))

    # ---- inference: one call per video -------------------------------------------------------------------------
    with torch.no_grad():
        tokens, info = arch.video_memory_tokens(model, frames, idx, mp, fp, model.image_newline)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            arch.video_memory_tokens(model, frames, idx, mp, fp, model.image_newline)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(f"inference : tokens {tuple(tokens.shape)}  memories {info['num_memories']}  {dt * 1e3:.2f} ms/video "
          f"({T / dt:,.0f} frames/s)")

    # ---- throughput mode: two videos in flight over one set of weights -----------------------------------------
    with torch.no_grad():
        pool = arch.MemoryPathPool(model, 2)
        pool.run([(frames, idx), (frames, idx)], mp, fp, model.image_newline)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pool.run([(frames, idx), (frames, idx)], mp, fp, model.image_newline)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / 20
    print(f"2 in flight: {dt2 * 1e3:.2f} ms/video ({T / dt2:,.0f} frames/s)")

    # ---- row batches: 8 videos stepped together per stream (their memory rows stacked into every weight-shared GEMM) ----
    with torch.no_grad():
        vids = [(torch.randn_like(frames) * 0.5, idx) for _ in range(16)]
        pool8 = arch.MemoryPathPool(model, 2, batch=8)
        for _ in range(5):                                    # (engines, workspaces and clocks warm: bench.py measures properly)
            pool8.run(vids, mp, fp, model.image_newline)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            pool8.run(vids, mp, fp, model.image_newline)
        torch.cuda.synchronize()
        dt8 = (time.perf_counter() - t0) / (20 * len(vids))
    print(f"2 streams x row batch of 8: {dt8 * 1e3:.3f} ms/video ({T / dt8:,.0f} frames/s)")

    # ---- hipGraph: the whole per-video launch sequence as one graph ---------------------------------------------
    g = arch.GraphedVideoMemory(model, T, idx)
    out = g(frames, mp.detach(), fp.detach(), model.image_newline)
    torch.cuda.synchronize()
    print(f"hipGraph  : replay equals eager launches: {torch.equal(out, tokens)}")

    # ---- training: same call with autograd enabled; forward and backward are HIP kernels ------------------------
    model.train()
    opt = torch.optim.SGD([p for n, p in model.named_parameters() if not n.startswith("embed_tokens")], lr=1e-4)
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        toks, _ = arch.video_memory_tokens(model, frames, idx, mp.detach(), fp.detach(), model.image_newline)
        loss = toks.float().square().mean()
        loss.backward()
        opt.step()
        print(f"train step {step}: loss {loss.item():.6f}  grad-norm(initial_memory) "
              f"{model.recurrent_memory_transformer.initial_memory.grad.float().norm().item():.3e}")


if __name__ == "__main__":
    main()
