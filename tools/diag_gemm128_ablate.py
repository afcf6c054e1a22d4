"""Diagnostic: where does the K loop of gemm128_kernel (128x256 tiles, two workgroups per CU) spend its time?
`build` (build container): textual ablations of the product source (never in the product build) -> lib/exp/libmavlm_g128_<v>.so.
`run` (GPU box): times every variant on two shapes, interleaved rounds.  Results of the ablated builds are WRONG by design.
usage: python tools/diag_gemm128_ablate.py build | run"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "memory-augmented-vlm_amd")
EXP = os.path.join(PKG, "lib", "exp")
VARIANTS = ["base", "nobdma", "noadma", "nodma", "noreads", "nomfma", "nobar", "nosetprio"]


def loop_only(src, fn):
    """apply fn to the text of the K loop only"""
    i = src.index("for (int kt = 0; kt < nk; ++kt) {")
    j = src.index("#undef MAVLM_QUAD")
    return src[:i] + fn(src[i:j]) + src[j:]


def make(src, v):
    if v == "base":
        return src
    if v in ("nobdma", "nodma"):
        src = loop_only(src, lambda t: re.sub(r"dma_b\(s[01], [01], kt \+ [12]\);", ";", t))
    if v in ("noadma", "nodma"):
        src = loop_only(src, lambda t: re.sub(r"dma_a\(s[01], [01], kt \+ [12]\);", ";", t))
    if v == "noreads":
        src = loop_only(src, lambda t: re.sub(r"(read_[ab]\()", r"if (kt == 0) \1", t))
    if v == "nomfma":
        src = src.replace("acc[MH * 4 + mt][NH * 2 + nt] = T::mfma16(bf[NH * 2 + nt][ks], AF[mt][ks], acc[MH * 4 + mt][NH * 2 + nt]);",
                          'asm volatile("" : "+v"(acc[MH * 4 + mt][NH * 2 + nt]) : "v"(bf[NH * 2 + nt][ks]), "v"(AF[mt][ks]));')
    if v == "nobar":
        src = loop_only(src, lambda t: t.replace("MAVLM_BAR1();", ";"))
    if v == "nosetprio":
        src = src.replace("__builtin_amdgcn_s_setprio(1);", ";").replace("__builtin_amdgcn_s_setprio(0);", ";")
    return src


def build():
    os.makedirs(EXP, exist_ok=True)
    src = open(os.path.join(PKG, "csrc", "gemm128.hip")).read()
    objs = [os.path.join(PKG, "lib", "obj", f) for f in os.listdir(os.path.join(PKG, "lib", "obj")) if f.endswith(".o") and f != "gemm128.o"]
    procs = []
    for v in VARIANTS:
        os.makedirs("/tmp/g128abl", exist_ok=True)
        p = f"/tmp/g128abl/g128_{v}.hip"
        open(p, "w").write(make(src, v))
        procs.append((v, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(PKG, "csrc"),
                                           "-c", "-o", f"/tmp/g128_{v}.o", p])))
    for v, pr in procs:
        assert pr.wait() == 0, v
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(EXP, f"libmavlm_g128_{v}.so"),
                        f"/tmp/g128_{v}.o"] + objs, check=True)
        print("built", v, flush=True)


def run_one():
    sys.path.insert(0, ROOT)
    import torch
    import memory_augmented_vlm_amd  # noqa: F401
    from memory_augmented_vlm_amd import _capi as capi, _ops as ops
    lib = capi.lib()
    assert lib.mavlm_set_gemm_tile(129) == 0
    for (M, N, K) in [(50176, 1024, 4096), (25088, 4096, 1024)]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16() * 0.05
        b32 = torch.randn(N, device="cuda")
        out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        for _ in range(20):
            ops.linear(a, w, b32, capi.EPI_BIAS, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.linear(a, w, b32, capi.EPI_BIAS, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        ts.sort()
        print(f"{os.environ.get('G128_V', '?'):10s} M{M} N{N} K{K}: {ts[2]:7.1f} us  ({2.0*M*N*K/ts[2]/1e6:7.1f} TF-equivalent)", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "one":
        run_one()
    else:
        for rnd in range(2):
            for v in VARIANTS:
                env = dict(os.environ, MAVLM_LIB=os.path.join(EXP, f"libmavlm_g128_{v}.so"), G128_V=v)
                subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, check=False)
