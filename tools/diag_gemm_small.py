"""Diagnostic: GEMM timings at the reference-default row count (M = 8 memory tokens -> 1568 rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
shapes = [(1568, 1024, 1024, 0), (1568, 4096, 1024, 1), (1568, 1024, 4096, 4), (6272, 4096, 1024, 0), (1568, 2048, 1024, 0),
          (784, 1024, 1024, 0), (392, 1024, 1024, 0)]
for tile in (0, 128, 256):
    capi.lib().mavlm_set_gemm_tile(tile)
    for (M, N, K, epi) in shapes:
        a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16(); b = torch.zeros(N, device="cuda")
        out = torch.empty((M, N), device="cuda", dtype=torch.float32 if epi == 4 else torch.bfloat16)
        import ctypes
        for _ in range(5): ops.linear(a, w, b, epi, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): ops.linear(a, w, b, epi, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        lib = capi.lib(); nk = len(capi.KERNEL_KINDS)
        ms = (ctypes.c_double * nk)(); ln = (ctypes.c_int64 * nk)(); fl = (ctypes.c_double * nk)(); by = (ctypes.c_double * nk)()
        lib.mavlm_prof_enable(1)
        for _ in range(50): ops.linear(a, w, b, epi, out=out)
        torch.cuda.synchronize()
        lib.mavlm_prof_read(ms, ln, fl, by, nk); lib.mavlm_prof_enable(0)
        ke = ms[0] / ln[0] * 1e3
        print(f"tile {tile:4d}  M{M:5d} N{N:5d} K{K:5d} epi{epi}: wall {dt*1e6:6.1f} us   kernel (HIP events) {ke:6.1f} us  {2*M*N*K/ke/1e6:6.1f} TF")
capi.lib().mavlm_set_gemm_tile(0)
