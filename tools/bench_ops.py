"""Micro-benchmark of the individual kernels at the bench shapes (not a test; used for A/B timing and for
rocprofv3 --pmc runs).  usage: python tools/bench_ops.py [attn|gemm|colsum|all] [iters]"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

what = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda"
torch.manual_seed(0)


def timeit(fn, flops, name, rounds=5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / iters)
    t.sort()
    ms, best = t[len(t) // 2], t[0]
    print(f"{name:44s} median {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s | best {best*1e3:8.1f} us {flops/best/1e9:7.1f} TF/s", flush=True)


R, S, H, D = 12544, 6272, 8, 1024
if what in ("attn", "all"):
  for impl in ((3,) if os.environ.get("ATTN_ONLY3") else (2, 3)):
    capi.lib().mavlm_set_attention_impl(impl)
    print("attention impl", impl)
    q = torch.randn(R, D, device=dev).bfloat16()
    kv = torch.randn(S, 4 * D, device=dev).bfloat16()
    timeit(lambda: ops.attention(q, kv[:, :D], kv[:, D:2 * D], H, want_lse=True), 4.0 * R * S * D, f"attn R={R} S={S}")
    kv2 = torch.randn(R, 2 * D, device=dev).bfloat16()
    timeit(lambda: ops.attention(q, kv2[:, :D], kv2[:, D:], H), 4.0 * R * R * D, f"attn R={R} S={R} (evolution n=1)")
  capi.lib().mavlm_set_attention_impl(0)
if what in ("colsum", "all"):
  for impl in (2, 3):
    capi.lib().mavlm_set_attention_impl(impl)
    q = torch.randn(R, D, device=dev).bfloat16()
    kv = torch.randn(S, 4 * D, device=dev).bfloat16()
    _, lse = ops.attention(q, kv[:, :D], kv[:, D:2 * D], H, want_lse=True)
    timeit(lambda: ops.attention_colsum(q, kv[:, :D], lse, H), 2.0 * R * S * D, f"colsum impl {impl} R={R} S={S}")
  capi.lib().mavlm_set_attention_impl(0)
if what in ("gemm", "all"):
  for tile, rows in ((256, 256), (256, 224), (257, 256), (257, 224), (0, 0)):
    capi.lib().mavlm_set_gemm_tile(tile)
    capi.lib().mavlm_set_gemm_rows(rows)
    print("tile", tile, "rows", rows)
    for (M, N, K, epi) in [(S, 4 * D, D, 0), (R, D, D, 0), (R, D, D, 4), (R, 4 * D, D, 1), (R, D, 4 * D, 4), (R, 2 * D, D, 0),
                           (R, 4 * D, D, 2)]:
      a = torch.randn(M, K, device=dev).bfloat16()
      w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
      b = torch.randn(N, device=dev)
      res = torch.randn(M, N, device=dev).bfloat16() if epi == 3 else None
      out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi in (3, 4) else torch.bfloat16)
      timeit(lambda: ops.linear(a, w, b, epi, residual=res, out=out), 2.0 * M * N * K, f"gemm M={M} N={N} K={K} epi={epi}")
if what in ("wide",):
    Hh, hd = 8, 448
    for (Rr, Ss) in ((1568, 6272), (12544, 6272)):
        q = torch.randn(Rr, Hh * hd, device=dev).bfloat16()
        kv = torch.randn(Ss, 2 * Hh * hd, device=dev).bfloat16()
        timeit(lambda: ops.attention(q, kv[:, :Hh * hd], kv[:, Hh * hd:], Hh, want_lse=True, head_dim=hd), 4.0 * Rr * Ss * Hh * hd,
               f"attn_hd448 R={Rr} S={Ss}")
        _, lse = ops.attention(q, kv[:, :Hh * hd], kv[:, Hh * hd:], Hh, want_lse=True, head_dim=hd)
        timeit(lambda: ops.attention_colsum(q, kv[:, :Hh * hd], lse, Hh, head_dim=hd), 2.0 * Rr * Ss * Hh * hd,
               f"colsum_hd448 R={Rr} S={Ss}")
    for (M, N, K, epi) in [(6272, 4 * 3584, 3584, 0), (1568, 3584, 3584, 4), (1568, 4 * 3584, 3584, 1), (1568, 3584, 4 * 3584, 4),
                           (12544, 4 * 3584, 3584, 1)]:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == 4 else torch.bfloat16)
        timeit(lambda: ops.linear(a, w, b, epi, out=out), 2.0 * M * N * K, f"gemm M={M} N={N} K={K} epi={epi}")
