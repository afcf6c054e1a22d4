"""Diagnostic: this library's GEMM against the vendor library (torch.nn.functional.linear -> hipBLASLt) at the bench
shapes, same box, bf16, HIP-event timing.  Evidence only - the product never calls hipBLASLt."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

def timeit_pair(fa, fb, n=30, rounds=5):
    """Interleaved A/B timing: both candidates are warmed up for ~0.25 s first (the first kernel measured after an idle
    gap runs 10-20 % slower while the clocks ramp - an ordering bias, not a property of either candidate), then `rounds`
    alternating blocks of n launches each; medians."""
    t0 = time.time()
    while time.time() - t0 < 0.25:
        for _ in range(10): fa()
        for _ in range(10): fb()
        torch.cuda.synchronize()
    ta, tb = [], []
    for _ in range(rounds):
        for f, acc in ((fa, ta), (fb, tb)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
            acc.append(e0.elapsed_time(e1) / n * 1e-3)
    ta.sort(); tb.sort()
    return ta[len(ta) // 2], tb[len(tb) // 2]

for (M, N, K) in [(12544, 1024, 1024), (12544, 4096, 1024), (12544, 1024, 4096), (6272, 4096, 1024), (12544, 2048, 1024),
                  (1568, 1024, 1024), (1568, 4096, 1024), (1568, 1024, 4096), (1568, 3584, 3584)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    b32 = torch.zeros(N, device="cuda"); b16 = b32.bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    t_mine, t_lt = timeit_pair(lambda: ops.linear(a, w, b32, capi.EPI_BIAS, out=out), lambda: torch.nn.functional.linear(a, w, b16))
    f = 2.0 * M * N * K
    print(f"M{M:6d} N{N:5d} K{K:5d}: this library {t_mine*1e6:7.1f} us {f/t_mine/1e12:7.1f} TF | hipBLASLt {t_lt*1e6:7.1f} us {f/t_lt/1e12:7.1f} TF | ratio {t_lt/t_mine:.2f}")

# attention: this library against torch's scaled_dot_product_attention (the ROCm flash / mem-efficient backends), same box
import torch.nn.functional as F_
for (R, S, H) in [(12544, 6272, 8), (12544, 12544, 8), (1568, 6272, 8)]:
    q = torch.randn(R, H * 128, device="cuda").bfloat16(); k = torch.randn(S, H * 128, device="cuda").bfloat16(); v = torch.randn(S, H * 128, device="cuda").bfloat16()
    q4 = q.view(R, H, 128).permute(1, 0, 2)[None].contiguous(); k4 = k.view(S, H, 128).permute(1, 0, 2)[None].contiguous(); v4 = v.view(S, H, 128).permute(1, 0, 2)[None].contiguous()
    try:
        t_mine, t_sdpa = timeit_pair(lambda: ops.attention(q, k, v, H, want_lse=True), lambda: F_.scaled_dot_product_attention(q4, k4, v4), 20, 3)
        msg = f"torch SDPA {t_sdpa*1e6:7.1f} us {4.0*R*S*H*128/t_sdpa/1e12:7.1f} TF | ratio {t_sdpa/t_mine:.2f}"
    except Exception as e:
        t_mine, _ = timeit_pair(lambda: ops.attention(q, k, v, H, want_lse=True), lambda: None, 20, 3)
        msg = f"torch SDPA unavailable ({type(e).__name__})"
    print(f"attention R{R} S{S} H{H}: this library {t_mine*1e6:7.1f} us {4.0*R*S*H*128/t_mine/1e12:7.1f} TF | {msg}")
