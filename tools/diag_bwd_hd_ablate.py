"""Diagnostic: where does a tile of attn_bwd_hd_kernel (head_dim 448 backward) spend its time?
`build` (build container): textual ablations of the product source (never in the product build) -> lib/exp/libmavlm_bhd_<v>.so.
`run` (GPU box): times every variant (dQ launch and dK+dV launch separately), interleaved rounds.  Results of the ablated builds
are WRONG by design.
usage: python tools/diag_bwd_hd_ablate.py build | run"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "memory-augmented-vlm_amd")
EXP = os.path.join(PKG, "lib", "exp")
VARIANTS = ["base", "nodma", "noxch", "nov", "nosmfma", "noamfma", "nosread", "nozread", "nobar2"]


def sub1(src, old, new):
    assert old in src, old
    return src.replace(old, new)


def make(src, v):
    if v == "nodma":
        src = sub1(src, "if constexpr (i < NPW) dma_piece(yrs, yoff, ldy, ytile, N * TILE, WIC<(i < NPW ? i : 0)>{});", "if constexpr (false) {}")
        src = sub1(src, "else if constexpr (i < 2 * NPW) dma_piece(y2rs, y2off, ldy2, y2tile, (2 + N) * TILE, WIC<(i < 2 * NPW ? i - NPW : 0)>{});", "")
    if v == "noxch":
        i = src.index("      if constexpr (TWO) {\n        // exchange with the partner wave")
        j = src.index("      } else {\n        tt = mine;\n      }")
        src = src[:i] + "      if constexpr (TWO) { tt = mine; dp = mine; asm volatile(\"\" : \"+v\"(dp)); }\n      else { tt = mine; }\n" + src[j + len("      } else {\n        tt = mine;\n      }"):]
    if v == "nov":
        src = sub1(src, "        float p = __builtin_amdgcn_exp2f(tt[i] * c - l);", "        float p = tt[i] + l;")
    if v == "nosmfma":
        src = sub1(src, "mine = T::mfma32(__builtin_bit_cast(typename T::vec8, yfr[i]), xf[i], mine);",
                   'asm volatile("" : "+v"(mine) : "v"(yfr[i]), "v"(xf[i]));')
    if v == "noamfma":
        src = sub1(src, "acc[db] = T::mfma32(__builtin_bit_cast(typename T::vec8, both), ef[sx], acc[db]);",
                   'asm volatile("" : "+v"(acc[db]) : "v"(both), "v"(ef[sx]));')
        src = sub1(src, "acc2[db] = T::mfma32(__builtin_bit_cast(typename T::vec8, b2), pf[sx], acc2[db]);",
                   'asm volatile("" : "+v"(acc2[db]) : "v"(b2), "v"(pf[sx]));')
    if v == "nosread":
        src = sub1(src, "if constexpr (i + KPF < KS) yrd(WIC<(i + KPF < KS ? i + KPF : KS - 1)>{});", "if constexpr (i + KPF < KS) yfr[i + KPF < KS ? i + KPF : 0] = yfr[i];")
    if v == "nozread":
        src = sub1(src, "if constexpr (i + KPF < NA) zrd(WIC<(i + KPF < NA ? i + KPF : NA - 1)>{});",
                   "if constexpr (i + KPF < NA) { zlo[i + KPF < NA ? i + KPF : 0] = zlo[i]; zhi[i + KPF < NA ? i + KPF : 0] = zhi[i]; if constexpr (DKV) { z2lo[i + KPF < NA ? i + KPF : 0] = z2lo[i]; z2hi[i + KPF < NA ? i + KPF : 0] = z2hi[i]; } }")
    if v == "nobar2":
        i = src.index('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave\'s DMAs of tile t+1 have landed')
        j = src.index("  };\n\n  int t = 0;")
        src = src[:i] + src[j:]
    return src


def build():
    os.makedirs(EXP, exist_ok=True)
    src = open(os.path.join(PKG, "csrc", "attention_bwd_hd.hip")).read()
    objs = [os.path.join(PKG, "lib", "obj", f) for f in os.listdir(os.path.join(PKG, "lib", "obj")) if f.endswith(".o") and f != "attention_bwd_hd.o"]
    procs = []
    os.makedirs("/tmp/bhdabl", exist_ok=True)
    for v in VARIANTS:
        p = f"/tmp/bhdabl/bhd_{v}.hip"
        open(p, "w").write(make(src, v))
        procs.append((v, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(PKG, "csrc"),
                                           "-c", "-o", f"/tmp/bhdabl/bhd_{v}.o", p])))
    for v, pr in procs:
        assert pr.wait() == 0, v
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(EXP, f"libmavlm_bhd_{v}.so"),
                        f"/tmp/bhdabl/bhd_{v}.o"] + objs, check=True)
        print("built", v, flush=True)


def run_one():
    sys.path.insert(0, ROOT)
    import torch
    import memory_augmented_vlm_amd  # noqa: F401
    from memory_augmented_vlm_amd import _ops as ops
    hd, H, R, S = 448, 8, 1568, 1568 + 32 * 196
    W = H * hd
    torch.manual_seed(0)
    q = (torch.randn(R, W, device="cuda") * 0.5).bfloat16()
    k = (torch.randn(S, W, device="cuda") * 0.5).bfloat16()
    v = torch.randn(S, W, device="cuda").bfloat16()
    do = (torch.randn(R, W, device="cuda") * 0.5).bfloat16()
    o, lse = ops.attention(q, k, v, H, want_lse=True, head_dim=hd)
    scale = ops.attn_scale(hd)
    res = []
    for (ndq, ndkv) in ((True, False), (False, True)):
        f = lambda: ops.attention_bwd_hd(q, k, v, o, do, lse, H, hd, scale, need_dq=ndq, need_dk=ndkv, need_dv=ndkv)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        ts.sort()
        units = 3.0 if ndq else 4.0
        res.append(f"{'dQ' if ndq else 'dK+dV'} {ts[2]:7.1f} us ({units * 2.0 * R * S * H * hd / ts[2] / 1e6:6.1f} TF)")
    print(f"{os.environ.get('BHD_V', '?'):10s} R{R} S{S} H{H}: " + " | ".join(res), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "one":
        run_one()
    else:
        for rnd in range(2):
            for v in VARIANTS:
                env = dict(os.environ, MAVLM_LIB=os.path.join(EXP, f"libmavlm_bhd_{v}.so"), BHD_V=v)
                subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, check=False)
