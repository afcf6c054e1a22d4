"""Diagnostic: one video at an arbitrary shape under rocprofv3 (HIDDEN / M / T from env), eager, 1 in flight."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
bench.HIDDEN = int(os.environ.get("HIDDEN", "1024")); bench.MEM_TOKENS = int(os.environ.get("MEM_TOKENS", "8"))
T = int(os.environ.get("FRAMES", "64"))
dev = torch.device("cuda", 0)
model, arch = bench.build_model(dev)
x = torch.randn(T, 196, bench.HIDDEN, device=dev).bfloat16(); idx = torch.arange(T) % 600
mp = torch.randn(10, bench.HIDDEN, device=dev).bfloat16(); fp = torch.randn(9, bench.HIDDEN, device=dev).bfloat16()
with torch.no_grad():
    for _ in range(6): arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
torch.cuda.synchronize()
