"""Diagnostic (GPU box): single-video latency of the path (one video, one stream) with the Residual blocks as one kernel /
as GEMM + LayerNorm kernel; exchange status of the fused form."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from memory_augmented_vlm_amd import _capi as capi  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model, arch = bench.build_model(dev)
x = torch.randn((64, 196, 1024)).to(dev).to(torch.bfloat16)
idx = torch.arange(64)
mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev)
frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
lib = capi.lib()


def step():
    mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
    fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
    return arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)[0]


with torch.no_grad():
    for rnd in range(3):
        for on in (1, 0):
            lib.mavlm_set_fused_layernorm(on)
            model.recurrent_memory_transformer._engine = None      # a context snapshots the hook when it is created
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            print(f"fused_layernorm={on}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per video; exchange status "
                  f"{model.recurrent_memory_transformer._engine.ln_exchange_status()}", flush=True)
    lib.mavlm_set_fused_layernorm(1)
    lib.mavlm_prof_enable(1)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    kern, ms, ln, fl, by = bench.kernel_table(lib, capi, 5)
    lib.mavlm_prof_enable(0)
    for k, v in kern.items():
        print(f"   {k:22s} {v['launches_per_step']:6.1f} launches  {v['avg_ms'] * 1e3:8.1f} us avg  {v['ms_per_step']:8.3f} ms/step  {v['tflops'] or 0:7.1f} TF")
