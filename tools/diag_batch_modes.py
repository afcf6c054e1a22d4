"""Diagnostic (GPU box): frames/s of the memory path for (streams, row batch) combinations, same process, interleaved
blocks (rule 24).  usage: python tools/diag_batch_modes.py [M] [configs...]   e.g.  64 2x1 1x2 2x2    /    8 2x1 1x8 2x8 2x4
env HIDDEN (1024) / FRAMES (64): e.g. HIDDEN=3584 FRAMES=256 ... 8 1x1 1x4 1x8 = BASELINE.json configs[2] (OV-7B width)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cfgs = [tuple(int(v) for v in c.split("x")) for c in (sys.argv[2:] or ["2x1", "1x2", "2x2"])]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    bench.HIDDEN = int(os.environ.get("HIDDEN", bench.HIDDEN))
    bench.FRAMES = int(os.environ.get("FRAMES", bench.FRAMES))
    model, arch = bench.build_model(dev, hidden=bench.HIDDEN, mem_tokens=M, seed=4321 if M == 8 else 1234)
    idx = torch.arange(bench.FRAMES)
    g = torch.Generator(device="cpu").manual_seed(100)
    nmax = max(s * b for s, b in cfgs)
    xs = [torch.randn((bench.FRAMES, bench.PATCHES, bench.HIDDEN), generator=g).to(dev).to(torch.bfloat16) for _ in range(nmax)]
    mem_ids = torch.tensor(arch.MEMORY_PROMPT_IDS, device=dev)
    frame_ids = torch.tensor(arch.FRAME_PROMPT_IDS, device=dev)
    pools = {c: arch.MemoryPathPool(model, c[0], batch=c[1]) for c in cfgs}

    def step(c):
        mp = torch.nn.functional.embedding(mem_ids, model.embed_tokens.weight)
        fp = torch.nn.functional.embedding(frame_ids, model.embed_tokens.weight)
        n = c[0] * c[1]
        return pools[c].run([(x, idx) for x in xs[:n]], mp, fp, model.image_newline)

    steps = int(os.environ.get("STEPS", "20" if bench.HIDDEN <= 1024 else "4"))
    res = {c: [] for c in cfgs}
    with torch.no_grad():
        for c in cfgs:
            for _ in range(3):
                step(c)
        torch.cuda.synchronize()
        for rnd in range(5):
            for c in cfgs:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step(c)
                torch.cuda.synchronize()
                res[c].append((time.perf_counter() - t0) / steps)
    fl = bench.algorithmic_flops(M=M, frames=bench.FRAMES, D=bench.HIDDEN)
    for c in cfgs:
        ts = sorted(res[c])
        med = ts[len(ts) // 2]
        n = c[0] * c[1]
        print(f"M={M} streams x batch = {c[0]}x{c[1]}: {n * bench.FRAMES / med:9.0f} frames/s  ({med * 1e3:.3f} ms per {n} videos, "
              f"path frac {n * fl / med / 1e12 / 2500:.3f}; min {n * bench.FRAMES / ts[-1]:.0f} max {n * bench.FRAMES / ts[0]:.0f})", flush=True)
    # per-kernel table of the last configuration (instrumented, one pass)
    from memory_augmented_vlm_amd import _capi as capi
    lib = capi.lib()
    for c in cfgs:
        with torch.no_grad():
            lib.mavlm_prof_enable(1)
            for _ in range(5):
                step(c)
            torch.cuda.synchronize()
            kern, ms, ln, flp, by = bench.kernel_table(lib, capi, 5)
            lib.mavlm_prof_enable(0)
        print(f"--- kernels, {c[0]}x{c[1]} (per step of {c[0] * c[1]} videos; instrumented: streams serialised by the event brackets)")
        for k, v in kern.items():
            print(f"   {k:22s} {v['launches_per_step']:6.1f} launches  {v['avg_ms'] * 1e3:8.1f} us avg  {v['ms_per_step']:8.3f} ms/step  "
                  f"{v['tflops'] or 0:7.1f} TF  {v['alg_gbs']:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
