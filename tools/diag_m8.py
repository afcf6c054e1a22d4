"""Diagnostic: throughput at the reference-default shape (M = 8 memory tokens) - eager / 2-3 videos in flight / hipGraph."""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if os.environ.get("HIDDEN"):
    bench.HIDDEN = int(os.environ["HIDDEN"])
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bench.MEM_TOKENS = M
dev = torch.device("cuda", 0)
model, arch = bench.build_model(dev, mem_tokens=M)
if os.environ.get("MAVLM_GEMM_TILE"):
    from memory_augmented_vlm_amd import _capi as _c
    _c.check(_c.lib().mavlm_set_gemm_tile(int(os.environ["MAVLM_GEMM_TILE"])), "tile")
x = torch.randn(T, 196, bench.HIDDEN, device=dev).bfloat16()
idx = torch.arange(T) % 600
mp = torch.randn(10, bench.HIDDEN, device=dev).bfloat16(); fp = torch.randn(9, bench.HIDDEN, device=dev).bfloat16()

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

with torch.no_grad():
    t1 = timed(lambda: arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline))
    print(f"M={M} T={T} eager, 1 in flight : {t1*1e3:7.3f} ms/video  {T/t1:9.0f} frames/s")
    for n in (2, 3):
        pool = arch.MemoryPathPool(model, n)
        tn = timed(lambda: pool.run([(x, idx)] * n, mp, fp, model.image_newline)) / n
        print(f"M={M} T={T} eager, {n} in flight : {tn*1e3:7.3f} ms/video  {T/tn:9.0f} frames/s")
    g = arch.GraphedVideoMemory(model, T, idx)
    tg = timed(lambda: g(x, mp, fp, model.image_newline))
    print(f"M={M} T={T} hipGraph, 1 in flight: {tg*1e3:7.3f} ms/video  {T/tg:9.0f} frames/s")
    for ng in (2, 3, 4, 6):
        gs = [arch.GraphedVideoMemory(model, T, idx) for _ in range(ng)]
        streams = [torch.cuda.Stream() for _ in gs]
        def many():
            for gg, st in zip(gs, streams):
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    gg(x, mp, fp, model.image_newline)
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
        tg = timed(many) / ng
        print(f"M={M} T={T} hipGraph, {ng} in flight: {tg*1e3:7.3f} ms/video  {T/tg:9.0f} frames/s")

# per-kernel-kind breakdown of one video (HIP events around every launch)
import ctypes
from memory_augmented_vlm_amd import _capi as capi
lib = capi.lib(); nk = len(capi.KERNEL_KINDS)
ms = (ctypes.c_double * nk)(); ln = (ctypes.c_int64 * nk)(); fl = (ctypes.c_double * nk)(); by = (ctypes.c_double * nk)()
with torch.no_grad():
    lib.mavlm_prof_enable(1)
    for _ in range(10):
        arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    capi.check(lib.mavlm_prof_read(ms, ln, fl, by, nk), "prof"); lib.mavlm_prof_enable(0)
for i, name in enumerate(capi.KERNEL_KINDS):
    if ln[i]:
        print(f"{name:18s} launches/video {ln[i]/10:5.1f}  ms/video {ms[i]/10:7.3f}  avg {ms[i]/ln[i]*1e3:7.1f} us"
              + (f"  {fl[i]/(ms[i]*1e-3)/1e12:6.1f} TF" if fl[i] else ""))
print(f"sum {sum(ms)/10:.3f} ms/video")
