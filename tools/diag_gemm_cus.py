"""Diagnostic: is the 256x256x64 GEMM loop bound per CU or chip-wide?  N = 1024, K = 4096, M = 256 * t/4 so that
t = 64 / 128 / 196 / 256 workgroups are active (one per CU); time per K-tile and the L2->LDS traffic it implies
(64 KiB per workgroup per K-tile)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

capi.lib().mavlm_set_gemm_tile(256)
capi.lib().mavlm_set_gemm_rows(256)
torch.manual_seed(0)
N = 1024
for K in (4096, 1024):
    for tiles in (32, 64, 128, 196, 224, 256):
        M = 256 * tiles // 4
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(10):
            ops.linear(a, w, b, 0, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.linear(a, w, b, 0, out=out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        ms = sorted(ts)[2]
        kt = K // 64
        print(f"K{K} workgroups {tiles:4d}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF  {ms * 1e3 / kt:.3f} us/K-tile "
              f"{tiles * 65536 * kt / (ms * 1e-3) / 1e12:5.2f} TB/s staged", flush=True)
