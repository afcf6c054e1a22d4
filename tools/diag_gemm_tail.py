"""Diagnostic (GPU box): would the LAST, partial round of a 256-row GEMM grid run faster as 128x256 tiles (gemm128.hip)?  Times the
tail of the 1.53- / 3.06-round shapes of the bench (the rows past the last full round of 256 tiles) with both kernels, and the
full-round part alone, against the whole launch.  usage: python tools/diag_gemm_tail.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from diag_vs_hipblaslt_util import timeit_pair

lib = capi.lib()
# (whole M, N, K): tiles = ceil(M/256) * N/256; full rounds of 256 tiles = rows [0, Mfull), tail = rows [Mfull, M)
for (M, N, K) in [(25088, 1024, 4096), (25088, 1024, 1024), (12544, 2048, 1024), (25088, 2048, 1024), (25088, 4096, 1024)]:
    ntn = N // 256
    tiles = -(-M // 256) * ntn
    full_tiles = (tiles // 256) * 256
    Mfull = (full_tiles // ntn) * 256
    Mtail = M - Mfull
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)

    def run(tile, r0, r1):
        def f():
            lib.mavlm_set_gemm_tile(tile)
            ops.linear(a[r0:r1], w, b32, capi.EPI_BIAS, out=out[r0:r1])
        return f
    def both():
        lib.mavlm_set_gemm_tile(256)
        ops.linear(a[:Mfull], w, b32, capi.EPI_BIAS, out=out[:Mfull])
        lib.mavlm_set_gemm_tile(129)
        ops.linear(a[Mfull:], w, b32, capi.EPI_BIAS, out=out[Mfull:])
    t_whole, t_full = timeit_pair(run(0, 0, M), run(256, 0, Mfull))
    t_tail256, t_tail128 = timeit_pair(run(256, Mfull, M), run(129, Mfull, M))
    t_both, t_whole2 = timeit_pair(both, run(0, 0, M))
    lib.mavlm_set_gemm_tile(0)
    t_whole, t_full, t_tail256, t_tail128, t_both, t_whole2 = (x * 1e6 for x in (t_whole, t_full, t_tail256, t_tail128, t_both, t_whole2))
    print(f"M {M} N {N} K {K}: {tiles} tiles, tail {Mtail} rows = {tiles - full_tiles} tiles | whole {t_whole:7.1f} us ({t_whole2:7.1f}) | full rounds {t_full:7.1f} | "
          f"tail: 256-row {t_tail256:7.1f}  128x256 {t_tail128:7.1f} | full rounds + 128x256 tail back to back {t_both:7.1f} ({t_both / t_whole2:.3f}x)", flush=True)
