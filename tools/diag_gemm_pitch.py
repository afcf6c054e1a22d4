"""Diagnostic (GPU box): does the row pitch of the GEMM operands matter?  Every operand of the path has a power-of-two pitch
(K = 1024 / 4096 elements); if the L2 channel of an address is taken from low address bits, the 256 rows of a K-slice a
workgroup stages fall on few channels.  Times the product GEMM with A / W stored with pitch K + pad elements.
usage: python tools/diag_gemm_pitch.py [tile]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from diag_vs_hipblaslt_util import timeit_pair

lib = capi.lib()
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for (M, N, K) in [(25088, 1024, 1024), (25088, 4096, 1024), (25088, 1024, 4096), (25088, 2048, 1024)]:
    a0 = torch.randn(M, K, device="cuda").bfloat16()
    w0 = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ref = None
    for (pa, pw) in [(0, 0), (64, 0), (0, 64), (64, 64), (128, 128), (192, 192), (32, 32), (8, 8)]:
        a = torch.zeros(M, K + pa, device="cuda", dtype=torch.bfloat16)[:, :K]
        w = torch.zeros(N, K + pw, device="cuda", dtype=torch.bfloat16)[:, :K]
        a.copy_(a0)
        w.copy_(w0)

        def run():
            lib.mavlm_set_gemm_tile(tile)
            ops.linear(a, w, b32, capi.EPI_BIAS, out=out)

        def base():
            lib.mavlm_set_gemm_tile(tile)
            ops.linear(a0, w0, b32, capi.EPI_BIAS, out=out)
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        ok = torch.equal(ref, out)
        t, tb = timeit_pair(run, base)
        f = 2.0 * M * N * K
        print(f"tile {tile} M{M} N{N} K{K} pad A {pa:3d} W {pw:3d}: {t*1e6:7.1f} us {f/t/1e12:7.1f} TF | unpadded {tb*1e6:7.1f} us "
              f"{f/tb/1e12:7.1f} TF | {tb/t:.3f}x {'ok' if ok else 'MISMATCH'}", flush=True)
lib.mavlm_set_gemm_tile(0)
