import os, sys, time
import torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
if os.environ.get("BIG_FIRST"):
    big, arch0 = bench.build_model(dev)                      # the M = 64 model of the main bench, used first
    xb = torch.randn(64, 196, 1024, device=dev).bfloat16()
    pb = arch0.MemoryPathPool(big, 2)
    with torch.no_grad():
        for _ in range(10):
            pb.run([(xb, torch.arange(64))] * 2, torch.zeros(10, 1024, device=dev).bfloat16(), torch.zeros(9, 1024, device=dev).bfloat16(), big.image_newline)
    torch.cuda.synchronize()
model, arch = bench.build_model(dev, mem_tokens=8, seed=4321)
g = torch.Generator(device="cpu").manual_seed(7)
xs = [torch.randn((64, 196, 1024), generator=g).to(dev).to(torch.bfloat16) for _ in range(2)]
idx = torch.arange(64)
pool = arch.MemoryPathPool(model, 2)
mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev); frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
mp0 = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight); fp0 = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
def timed(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    a = timed(lambda: pool.run([(xi, idx) for xi in xs], mp0, fp0, model.image_newline))
    def step():
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        return pool.run([(xi, idx) for xi in xs], mp, fp, model.image_newline)[0]
    b = timed(step)
    c = timed(lambda: pool.run([(xs[0], idx)] * 2, mp0, fp0, model.image_newline))
    print(f"two inputs, fixed prompts {a*1e3:.3f} ms/step; with embedding {b*1e3:.3f}; same input twice {c*1e3:.3f}")
    # host-only time: how long does the python side take when the GPU is not the bottleneck
    t0 = time.perf_counter()
    for _ in range(40): step()
    host = (time.perf_counter() - t0) / 40
    torch.cuda.synchronize()
    print(f"host enqueue time per step {host*1e3:.3f} ms")
