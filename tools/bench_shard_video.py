"""Diagnostic (not the headline bench): ONE video with its memory rows sharded over the ranks (RowShardedMemory,
SURVEY.md §8e option 2) - strong scaling of a single video's latency.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_shard_video.py [frames] [steps]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from memory_augmented_vlm_amd import distributed as D

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rank, world, local = D.init_from_env()
dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))
torch.cuda.set_device(dev)
model, arch = bench.build_model(dev)
rm = model.recurrent_memory_transformer
sh = D.RowShardedMemory(rm)
x = torch.randn(T, 196, bench.HIDDEN, device=dev).bfloat16()

def video():
    sh.reset()
    for i in range(0, T, 32):
        sh.step(x[i:i + 32])

for _ in range(3): video()
torch.cuda.synchronize()
if world > 1: dist.barrier()
t0 = time.perf_counter()
for _ in range(steps): video()
torch.cuda.synchronize()
if world > 1: dist.barrier()
dt = (time.perf_counter() - t0) / steps
if rank == 0:
    print(f"row-sharded video: {world} rank(s), {T} frames, M={bench.MEM_TOKENS}: {dt*1e3:.2f} ms/video ({T/dt:,.0f} frames/s)")
if world > 1: dist.destroy_process_group()
