"""Diagnostic (GPU box): single-video latency at the checkpoint shape (M = 8, D = 1024) with the frame scores on the forward's tile
entries (mode 1: automatic) against the column-sum pass (mode 0), and the headline shape (M = 64, 2 x 2) with the per-(row, frame)
form (mode 1) against the tile-entry form forced everywhere (mode 2); same process, interleaved.  usage: python tools/diag_m8_scores_ab.py"""
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
for (M, st, b, modes) in ((8, 1, 1, (1, 0)), (8, 2, 1, (1, 0)), (64, 2, 2, (1, 2)), (64, 1, 1, (1, 2)), (8, 2, 8, (1, 2)), (8, 1, 8, (1, 2))):
    model, arch = bench.build_model(dev, hidden=1024, mem_tokens=M, seed=4321 if M == 8 else 1234)
    idx = torch.arange(bench.FRAMES)
    g = torch.Generator(device="cpu").manual_seed(100)
    xs = [torch.randn((bench.FRAMES, bench.PATCHES, 1024), generator=g).to(dev).to(torch.bfloat16) for _ in range(st * b)]
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
    pool = arch.MemoryPathPool(model, st, batch=b)

    def step():
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        return pool.run([(x, idx) for x in xs], mp, fp, model.image_newline)
    res = {m: [] for m in modes}
    n = 40 if M == 8 else 10
    with torch.no_grad():
        for mode in modes:
            lib.mavlm_set_frame_score_mode(mode)
            for _ in range(5):
                step()
        for rnd in range(5):
            for mode in modes:
                lib.mavlm_set_frame_score_mode(mode)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    step()
                torch.cuda.synchronize(); res[mode].append((time.perf_counter() - t0) / n)
    lib.mavlm_set_frame_score_mode(1)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    print(f"M={M} {st} stream(s) x batch {b}: " + " | ".join(f"mode {m}: {med[m] * 1e3:8.3f} ms" for m in modes) +
          f" | {med[modes[1]] / med[modes[0]]:.4f}x", flush=True)
    del pool, model
