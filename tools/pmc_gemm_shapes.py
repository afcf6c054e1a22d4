"""Driven by tools/pmc_gemm.sh under rocprofv3 --pmc: every GEMM shape / epilogue of the path, a few launches each (no timing)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

EPI = {"bias": capi.EPI_BIAS, "relu": capi.EPI_RELU, "gelu": capi.EPI_GELU, "f32": capi.EPI_F32}
CASES = [(12544, 1024, 1024, "bias"), (12544, 4096, 1024, "relu"), (12544, 1024, 4096, "f32"), (12544, 2048, 1024, "bias"),
         (25088, 1024, 1024, "bias"), (25088, 4096, 1024, "relu"), (25088, 4096, 1024, "gelu"), (25088, 1024, 4096, "f32"),
         (25088, 2048, 1024, "bias"), (6272, 4096, 1024, "bias")]
for (M, N, K, e) in CASES:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if e == "f32" else torch.bfloat16)
    for _ in range(4):
        ops.linear(a, w, b32, EPI[e], out=out)
    torch.cuda.synchronize()
for (M, N, K) in [(25088, 1024, 1024), (25088, 1024, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").bfloat16()
    g = torch.rand(N, device="cuda") + 0.5
    be = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    for _ in range(4):
        ops.linear_residual_layernorm(a, w, b32, res, g, be, 1e-12, False, out)
    torch.cuda.synchronize()
