"""Diagnostic: the non-persistent 256-tile GEMM at a long-K and a short-K shape (in-loop vs fixed cost), used with the
ablation builds of tools/diag_gemm_ablate.sh (MAVLM_LIB=...)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

capi.lib().mavlm_set_gemm_tile(256)
torch.manual_seed(0)
EPI = int(os.environ.get("GEMM_EPI", "0"))          # 0 bias -> 16-bit, 4 bias -> fp32 (feeds the LayerNorm kernel)
for rows in (224, 256, 224, 256):
    capi.lib().mavlm_set_gemm_rows(rows)
    for (M, N, K) in [(12544, 1024, 4096), (12544, 1024, 1024), (12544, 1024, 8192)]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if EPI == 4 else torch.bfloat16)
        for _ in range(10):
            ops.linear(a, w, b, EPI, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.linear(a, w, b, EPI, out=out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        ms = sorted(ts)[2]
        print(f"epi {EPI} rows {rows} M{M} N{N} K{K}: {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:7.1f} TF  ({ms * 1e3 / (K // 64):.3f} us per K-tile)", flush=True)
