"""Diagnostic (GPU box): the 128x256 two-per-CU GEMM (hook 129) against the automatic choice at the SMALL row counts of a single
video with 8 memory tokens (R = 1568) - OneVision-7B width and D = 1024 - where the 256-row kernels do not fill the chip.
usage: python tools/diag_gemm128_small.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from diag_vs_hipblaslt_util import timeit_pair

lib = capi.lib()
EPI = {"bias": capi.EPI_BIAS, "relu": capi.EPI_RELU, "gelu": capi.EPI_GELU, "f32": capi.EPI_F32}
CASES = [(1568, 3584, 3584, "bias"), (1568, 3584, 3584, "f32"), (1568, 14336, 3584, "relu"), (1568, 3584, 14336, "f32"),
         (1568, 7168, 3584, "bias"), (6272, 14336, 3584, "bias"), (3136, 3584, 3584, "bias"), (6272, 3584, 3584, "bias"),
         (1568, 1024, 1024, "bias"), (1568, 4096, 1024, "relu"), (1568, 1024, 4096, "f32"), (1568, 2048, 1024, "bias"),
         (3136, 1024, 1024, "bias"), (6272, 1024, 1024, "bias"), (6272, 4096, 1024, "bias"), (15680, 4096, 3584 // 3584 * 1024, "gelu")]
for (M, N, K, e) in CASES:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    b16 = b32.bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if e == "f32" else torch.bfloat16)
    nws = int(lib.mavlm_linear_ws_floats(M, N, K, EPI[e], N))

    def new():
        lib.mavlm_set_gemm_tile(129)
        ops.linear(a, w, b32, EPI[e], out=out)

    def old():
        lib.mavlm_set_gemm_tile(0)
        ops.linear(a, w, b32, EPI[e], out=out)
    t_new, t_old = timeit_pair(new, old)
    lib.mavlm_set_gemm_tile(0)
    t_new2, t_lt = timeit_pair(new, lambda: torch.nn.functional.linear(a, w, b16))
    lib.mavlm_set_gemm_tile(0)
    f = 2.0 * M * N * K
    print(f"M{M:6d} N{N:6d} K{K:6d} {e:5s}: 128x256 2/CU {t_new*1e6:7.1f} us {f/t_new/1e12:7.1f} TF | automatic{' (split-K)' if nws else ''} "
          f"{t_old*1e6:7.1f} us {f/t_old/1e12:7.1f} TF ({t_old/t_new:.3f}x) | hipBLASLt(bias) {t_lt*1e6:7.1f} us {f/t_lt/1e12:7.1f} TF", flush=True)
