"""Diagnostic: per-K-tile cost vs per-tile overhead of the 256x256 GEMM kernels (K sweep at fixed M x N)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
dev = "cuda"
def t(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3
for (M, N) in ((12544, 4096), (12544, 1024), (6272, 4096)):
    for tile in (256, 257):
        capi.lib().mavlm_set_gemm_tile(tile)
        res = []
        for K in (128, 256, 512, 1024, 2048, 4096):
            a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
            b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            res.append((K, t(lambda: ops.linear(a, w, b, 0, out=out))))
        tiles = ((M + 255) // 256) * (N // 256)
        # least squares: time = c0 + c1 * (K/64)
        import numpy as np
        x = np.array([k / 64 for k, _ in res]); y = np.array([v for _, v in res])
        c1, c0 = np.polyfit(x, y, 1)
        print(f"M={M} N={N} tiles={tiles} kernel={tile}: " + " ".join(f"K{k}:{v:.1f}us" for k, v in res) + f" | fit: {c0:.1f} us + {c1:.2f} us/K-tile")
