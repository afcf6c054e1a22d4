"""Diagnostic: throughput of 1 vs 2 vs 3 independent videos in flight on separate HIP streams (same weights)."""
import copy, os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
model, arch = bench.build_model(dev)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
models = [model] + [copy.deepcopy(model) for _ in range(NS - 1)]
for m in models[1:]:
    m.recurrent_memory_transformer._engine = None
    m.recurrent_memory_transformer.bind_fuser(m.memory_fuser, m.token_type_embedding)
streams = [torch.cuda.Stream() for _ in range(NS)]
x = torch.randn(64, 196, 1024, device=dev).bfloat16()
idx = torch.arange(64)
mp = torch.randn(10, 1024, device=dev).bfloat16(); fp = torch.randn(9, 1024, device=dev).bfloat16()

def step():
    for m, s in zip(models, streams):
        with torch.cuda.stream(s):
            arch.video_memory_tokens(m, x, idx, mp, fp, m.image_newline)

with torch.no_grad():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 15
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{NS} streams: {NS*K*64/dt:.0f} frames/s, {dt/K/NS*1e3:.3f} ms per video")
