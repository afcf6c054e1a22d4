"""Diagnostic (GPU box): frame scores at the OneVision-7B width riding on the forward (mode 1) against the column-sum pass (mode 0),
same process, interleaved: one 256-frame video and a row batch of four.  usage: python tools/diag_wide_scores_ab.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from memory_augmented_vlm_amd import _capi as capi  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
bench.HIDDEN, bench.FRAMES = 3584, 256
model, arch = bench.build_model(dev, hidden=3584, mem_tokens=8, seed=4321)
idx = torch.arange(bench.FRAMES)
g = torch.Generator(device="cpu").manual_seed(100)
xs = [torch.randn((bench.FRAMES, bench.PATCHES, 3584), generator=g).to(dev).to(torch.bfloat16) for _ in range(4)]
mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev)
frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
lib = capi.lib()
for (st, b) in ((1, 1), (1, 4)):
    pool = arch.MemoryPathPool(model, st, batch=b)

    def step():
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        return pool.run([(x, idx) for x in xs[:st * b]], mp, fp, model.image_newline)
    res = {0: [], 1: []}
    with torch.no_grad():
        for mode in (1, 0):
            lib.mavlm_set_frame_score_mode(mode)
            for _ in range(2):
                step()
        for rnd in range(4):
            for mode in (1, 0):
                lib.mavlm_set_frame_score_mode(mode)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3):
                    step()
                torch.cuda.synchronize(); res[mode].append((time.perf_counter() - t0) / 3)
    lib.mavlm_set_frame_score_mode(1)
    m1, m0 = sorted(res[1])[len(res[1]) // 2], sorted(res[0])[len(res[0]) // 2]
    print(f"{st} stream x batch {b}: scores on the forward {m1 * 1e3:8.3f} ms | column-sum pass {m0 * 1e3:8.3f} ms | {m0 / m1:.4f}x", flush=True)
