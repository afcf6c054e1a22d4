"""Driven by tools/pmc_gemm_ab.sh under rocprofv3 --pmc: the 256-row kernels (tile 0) and the 128x256 two-per-CU kernel (129)
on the same shapes, a few launches each (no timing)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

lib = capi.lib()
for (M, N, K) in [(49152, 1024, 4096), (24576, 4096, 1024), (24576, 1024, 1024)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    for tile in (0, 129):
        lib.mavlm_set_gemm_tile(tile)
        for _ in range(4):
            ops.linear(a, w, b32, capi.EPI_BIAS, out=out)
        torch.cuda.synchronize()
lib.mavlm_set_gemm_tile(0)
