"""Diagnostic (GPU box): the GEMMs of the path shape by shape and epilogue by epilogue (single video: 12 544 rows; row batch of
two: 25 088) against hipBLASLt through torch (bias only - the vendor call has no ReLU / GELU / fp32 epilogue), same box,
interleaved.  usage: python tools/diag_gemm_shapes.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from diag_vs_hipblaslt_util import timeit_pair

EPI = {"bias": capi.EPI_BIAS, "relu": capi.EPI_RELU, "gelu": capi.EPI_GELU, "f32": capi.EPI_F32}
CASES = [(12544, 1024, 1024, "bias"), (12544, 1024, 1024, "f32"), (12544, 4096, 1024, "relu"), (12544, 4096, 1024, "gelu"),
         (12544, 1024, 4096, "f32"), (12544, 1024, 4096, "bias"), (6272, 4096, 1024, "bias"), (12544, 2048, 1024, "bias"),
         (25088, 1024, 1024, "bias"), (25088, 1024, 1024, "f32"), (25088, 4096, 1024, "relu"), (25088, 4096, 1024, "gelu"),
         (25088, 1024, 4096, "f32"), (25088, 1024, 4096, "bias"), (25088, 2048, 1024, "bias")]
for (M, N, K, e) in CASES:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    b16 = b32.bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if e == "f32" else torch.bfloat16)
    t_mine, t_lt = timeit_pair(lambda: ops.linear(a, w, b32, EPI[e], out=out), lambda: torch.nn.functional.linear(a, w, b16))
    f = 2.0 * M * N * K
    print(f"M{M:6d} N{N:5d} K{K:5d} {e:5s}: this library {t_mine*1e6:7.1f} us {f/t_mine/1e12:7.1f} TF | hipBLASLt(bias) "
          f"{t_lt*1e6:7.1f} us {f/t_lt/1e12:7.1f} TF | ratio {t_lt/t_mine:.2f}", flush=True)

# the Residual block: one kernel (mavlm_linear_ln) against GEMM (fp32 epilogue) + row LayerNorm kernel
lib = capi.lib()
for (M, N, K) in [(12544, 1024, 1024), (12544, 1024, 4096), (25088, 1024, 1024), (25088, 1024, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").bfloat16()
    g = torch.rand(N, device="cuda") + 0.5
    be = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)

    def fused():
        lib.mavlm_set_fused_layernorm(1)
        ops.linear_residual_layernorm(a, w, b32, res, g, be, 1e-12, out=out)

    def two():
        lib.mavlm_set_fused_layernorm(0)
        ops.linear_residual_layernorm(a, w, b32, res, g, be, 1e-12, out=out)
    t_f, t_2 = timeit_pair(fused, two)
    lib.mavlm_set_fused_layernorm(1)
    f = 2.0 * M * N * K
    print(f"Residual block M{M:6d} N{N:5d} K{K:5d}: one kernel {t_f*1e6:7.1f} us ({f/t_f/1e12:7.1f} TF) | GEMM + LayerNorm kernels "
          f"{t_2*1e6:7.1f} us | ratio {t_2/t_f:.2f}", flush=True)
