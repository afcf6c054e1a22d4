"""Diagnostic (GPU box): the 128x256 two-workgroups-per-CU GEMM (gemm128.hip, mavlm_set_gemm_tile(129)) against the
256-row kernels (automatic choice) shape by shape: bit-identity of the outputs, then interleaved timing, then hipBLASLt (bias only).
usage: python tools/diag_gemm128.py [quick]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
from diag_vs_hipblaslt_util import timeit_pair

lib = capi.lib()
assert lib.mavlm_set_gemm_tile(129) == 0 and lib.mavlm_set_gemm_tile(0) == 0
EPI = {"bias": capi.EPI_BIAS, "relu": capi.EPI_RELU, "gelu": capi.EPI_GELU, "f32": capi.EPI_F32}
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"

# ---- correctness: ragged M, every epilogue, both dtypes
bad = 0
for dt in (torch.bfloat16, torch.float16):
    for (M, N, K) in [(128, 256, 64), (100, 256, 128), (1568, 1024, 1024), (12544, 512, 192), (777, 768, 4096)]:
        for e in ("bias", "relu", "gelu", "f32"):
            a = torch.randn(M, K, device="cuda").to(dt)
            w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
            b32 = torch.randn(N, device="cuda")
            lib.mavlm_set_gemm_tile(256)
            ref = ops.linear(a, w, b32, EPI[e])
            lib.mavlm_set_gemm_tile(129)
            got = ops.linear(a, w, b32, EPI[e])
            lib.mavlm_set_gemm_tile(0)
            torch.cuda.synchronize()
            same = torch.equal(ref, got)
            if not same:
                bad += 1
                d = (ref.float() - got.float()).abs().max().item()
                print(f"MISMATCH {dt} M{M} N{N} K{K} {e}: max abs diff {d:.3e}", flush=True)
print("bit-identity vs the 256-row kernel:", "OK" if bad == 0 else f"{bad} mismatches", flush=True)

CASES = [(12544, 1024, 1024, "bias"), (12544, 4096, 1024, "relu"), (12544, 1024, 4096, "bias"), (6272, 4096, 1024, "bias"),
         (12544, 2048, 1024, "bias"),
         (25088, 1024, 1024, "bias"), (25088, 4096, 1024, "relu"), (25088, 4096, 1024, "gelu"), (25088, 1024, 4096, "bias"),
         (25088, 2048, 1024, "bias"), (50176, 4096, 1024, "gelu"), (50176, 1024, 4096, "bias")]
if quick:
    CASES = CASES[5:10]
for (M, N, K, e) in CASES:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    b16 = b32.bfloat16()
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if e == "f32" else torch.bfloat16)

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
    print(f"M{M:6d} N{N:5d} K{K:5d} {e:5s}: 128x256 2/CU {t_new*1e6:7.1f} us {f/t_new/1e12:7.1f} TF | 256-row {t_old*1e6:7.1f} us "
          f"{f/t_old/1e12:7.1f} TF ({t_old/t_new:.3f}x) | hipBLASLt(bias) {t_lt*1e6:7.1f} us {f/t_lt/1e12:7.1f} TF ({t_lt/t_new2:.3f}x)",
          flush=True)

# ---- the Residual block (EPI_LN)
for (M, N, K) in [(12544, 1024, 1024), (12544, 1024, 4096), (25088, 1024, 1024), (25088, 1024, 4096)]:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
    b32 = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").bfloat16()
    g = torch.rand(N, device="cuda") + 0.5
    be = torch.randn(N, device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    out2 = torch.empty_like(out)
    lib.mavlm_set_gemm_tile(129)
    ops.linear_residual_layernorm(a, w, b32, res, g, be, 1e-12, out=out)
    lib.mavlm_set_gemm_tile(0)
    ops.linear_residual_layernorm(a, w, b32, res, g, be, 1e-12, out=out2)
    torch.cuda.synchronize()
    d = (out.float() - out2.float()).abs().max().item()

    def new():
        lib.mavlm_set_gemm_tile(129)
        ops.linear_residual_layernorm(a, w, b32, res, g, be, 1e-12, out=out)

    def old():
        lib.mavlm_set_gemm_tile(0)
        ops.linear_residual_layernorm(a, w, b32, res, g, be, 1e-12, out=out)
    t_new, t_old = timeit_pair(new, old)
    lib.mavlm_set_gemm_tile(0)
    f = 2.0 * M * N * K
    print(f"Residual block M{M:6d} N{N:5d} K{K:5d}: 128x256 2/CU {t_new*1e6:7.1f} us ({f/t_new/1e12:7.1f} TF) | 256-row one kernel "
          f"{t_old*1e6:7.1f} us ({f/t_old/1e12:7.1f} TF) ({t_old/t_new:.3f}x) | max abs diff of the two {d:.2e}", flush=True)
