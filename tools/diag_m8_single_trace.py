"""Diagnostic (GPU box): 30 single videos at the checkpoint shape (M = 8, D = 1024, 64 frames), one stream - run it under
`rocprofv3 --kernel-trace --stats` to see the kernel sequence of ONE video (HISTORY R4.9).  usage: rocprofv3 ... -- python3 tools/diag_m8_single_trace.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
model, arch = bench.build_model(dev, hidden=1024, mem_tokens=8, seed=4321)
idx = torch.arange(64)
x = torch.randn((64, 196, 1024), device=dev).to(torch.bfloat16)
mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev); frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
pool = arch.MemoryPathPool(model, 1, batch=1)
with torch.no_grad():
    for _ in range(30):
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        pool.run([(x, idx)], mp, fp, model.image_newline)
torch.cuda.synchronize()
