"""Diagnostic (GPU box): the next chunk's K/V projection on a side stream beside the current step (llava_arch.PROJECT_AHEAD: True = on wherever the
memory rows are few, False = off), single-video latency, same process, interleaved.
usage: python tools/diag_ahead_ab.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for (hidden, frames, M, st) in ((1024, 64, 8, 1), (1024, 256, 8, 1), (3584, 256, 8, 1), (1024, 64, 8, 2)):
    bench.HIDDEN, bench.FRAMES = hidden, frames
    model, arch = bench.build_model(dev, hidden=hidden, mem_tokens=M, seed=4321)
    idx = torch.arange(frames)
    g = torch.Generator(device="cpu").manual_seed(100)
    xs = [torch.randn((frames, bench.PATCHES, hidden), generator=g).to(dev).to(torch.bfloat16) for _ in range(st)]
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
    pool = arch.MemoryPathPool(model, st, batch=1)

    def step():
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        return pool.run([(x, idx) for x in xs], mp, fp, model.image_newline)
    res = {True: [], False: []}
    n = 20 if hidden == 1024 else 3
    with torch.no_grad():
        for mode in (True, False):
            arch.PROJECT_AHEAD = mode
            for _ in range(3):
                step()
        for rnd in range(5):
            for mode in (True, False):
                arch.PROJECT_AHEAD = mode
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    step()
                torch.cuda.synchronize(); res[mode].append((time.perf_counter() - t0) / n)
    arch.PROJECT_AHEAD = None
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    print(f"D={hidden} frames={frames} M={M} {st} stream(s): ahead {med[True] * 1e3:8.3f} ms | off {med[False] * 1e3:8.3f} ms | {med[False] / med[True]:.4f}x", flush=True)
    del pool, model
