"""Diagnostic (GPU box), round 4: (1) tile order of the persistent GEMM (mavlm_set_gemm_order 1 vs 0) at the N >= 2048 shapes,
(2) the GELU epilogue (packed fp32 math) against the ReLU launch of the same shape.  Interleaved timing, same box.
usage: python tools/diag_gemm_r4.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from diag_vs_hipblaslt_util import timeit_pair

lib = capi.lib()
for (M, N, K, e) in [(25088, 4096, 1024, capi.EPI_RELU), (25088, 2048, 1024, capi.EPI_BIAS), (50176, 4096, 1024, capi.EPI_GELU),
                     (6272, 4096, 1024, capi.EPI_BIAS), (12544, 4096, 1024, capi.EPI_RELU), (25088, 1024, 4096, capi.EPI_BIAS)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)

    def new():
        lib.mavlm_set_gemm_order(1)
        ops.linear(a, w, b32, e, out=out)

    def old():
        lib.mavlm_set_gemm_order(0)
        ops.linear(a, w, b32, e, out=out)
    new(); r1 = out.clone(); old(); torch.cuda.synchronize()
    same = torch.equal(r1, out)
    t1, t0 = timeit_pair(new, old)
    lib.mavlm_set_gemm_order(1)
    f = 2.0 * M * N * K
    print(f"order M{M:6d} N{N:5d} K{K:5d} epi {e}: 8x4 blocks {t1*1e6:7.1f} us {f/t1/1e12:7.1f} TF | consecutive {t0*1e6:7.1f} us {f/t0/1e12:7.1f} TF "
          f"| {t0/t1:.3f}x {'same bits' if same else 'MISMATCH'}", flush=True)
for (M, N, K) in [(25088, 4096, 1024), (50176, 4096, 1024), (12544, 4096, 1024)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    tg, tr = timeit_pair(lambda: ops.linear(a, w, b32, capi.EPI_GELU, out=out), lambda: ops.linear(a, w, b32, capi.EPI_RELU, out=out))
    ref = torch.nn.functional.gelu(torch.nn.functional.linear(a.float(), w.float(), b32))
    ops.linear(a, w, b32, capi.EPI_GELU, out=out)
    err = ((out.float() - ref).norm() / ref.norm()).item()
    f = 2.0 * M * N * K
    print(f"gelu  M{M:6d} N{N:5d} K{K:5d}: GELU {tg*1e6:7.1f} us {f/tg/1e12:7.1f} TF | ReLU {tr*1e6:7.1f} us {f/tr/1e12:7.1f} TF | GELU/ReLU {tg/tr:.3f} "
          f"| rel-L2 vs torch fp32 {err:.2e}", flush=True)
