"""Diagnostic: this library's GEMM against the vendor library (torch.nn.functional.linear -> hipBLASLt) at the bench
shapes, same box, bf16, HIP-event timing.  Evidence only - the product never calls hipBLASLt."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for (M, N, K) in [(12544, 1024, 1024), (12544, 4096, 1024), (12544, 1024, 4096), (6272, 4096, 1024), (12544, 2048, 1024),
                  (1568, 1024, 1024), (1568, 4096, 1024), (1568, 1024, 4096), (1568, 3584, 3584)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    b32 = torch.zeros(N, device="cuda"); b16 = b32.bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    t_mine = timeit(lambda: ops.linear(a, w, b32, capi.EPI_BIAS, out=out))
    t_lt = timeit(lambda: torch.nn.functional.linear(a, w, b16))
    f = 2.0 * M * N * K
    print(f"M{M:6d} N{N:5d} K{K:5d}: this library {t_mine*1e6:7.1f} us {f/t_mine/1e12:7.1f} TF | hipBLASLt {t_lt*1e6:7.1f} us {f/t_lt/1e12:7.1f} TF | ratio {t_lt/t_mine:.2f}")

# attention: this library against torch's scaled_dot_product_attention (the ROCm flash / mem-efficient backends), same box
import torch.nn.functional as F_
for (R, S, H) in [(12544, 6272, 8), (12544, 12544, 8), (1568, 6272, 8)]:
    q = torch.randn(R, H * 128, device="cuda").bfloat16(); k = torch.randn(S, H * 128, device="cuda").bfloat16(); v = torch.randn(S, H * 128, device="cuda").bfloat16()
    t_mine = timeit(lambda: ops.attention(q, k, v, H, want_lse=True), 20)
    q4 = q.view(R, H, 128).permute(1, 0, 2)[None].contiguous(); k4 = k.view(S, H, 128).permute(1, 0, 2)[None].contiguous(); v4 = v.view(S, H, 128).permute(1, 0, 2)[None].contiguous()
    try:
        t_sdpa = timeit(lambda: F_.scaled_dot_product_attention(q4, k4, v4), 20)
        msg = f"torch SDPA {t_sdpa*1e6:7.1f} us {4.0*R*S*H*128/t_sdpa/1e12:7.1f} TF | ratio {t_sdpa/t_mine:.2f}"
    except Exception as e:
        msg = f"torch SDPA unavailable ({type(e).__name__})"
    print(f"attention R{R} S{S} H{H}: this library {t_mine*1e6:7.1f} us {4.0*R*S*H*128/t_mine/1e12:7.1f} TF | {msg}")
