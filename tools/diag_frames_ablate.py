"""Diagnostic: where does the frame-score variant of attn_fwd3_kernel (FR = 1) lose its 10-12 % against the plain launch?
`build` (build container): textual ablations of the product source -> lib/exp/libmavlm_fr_<v>.so; `run` (GPU box): times
ops.attention_frames against ops.attention(plain) at the bench's single-video and row-batch-of-2 shapes, interleaved rounds.
Results of the ablated builds are WRONG by design.  usage: python tools/diag_frames_ablate.py build | run"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "memory-augmented-vlm_amd")
EXP = os.path.join(PKG, "lib", "exp")
VARIANTS = ["base", "nostore", "nogroups", "noflush", "noreadback", "nothing"]


def sub1(src, old, new):
    assert src.count(old) == 1, (src.count(old), old)
    return src.replace(old, new)


def make(src, v):
    if v == "nostore":
        src = sub1(src, "          __builtin_amdgcn_raw_buffer_store_b64(e, frs, f_voff, f_cur * 8, 0);\n        }\n        a_cur = psum - plo;",
                   "          asm volatile(\"\" :: \"v\"(e));\n        }\n        a_cur = psum - plo;")
    if v == "nogroups":
        i = src.index("        float plo;\n        if (kofs <= 32) {                                     // boundary in block 0")
        j = src.index("        const float a_done = xhalf_sum(a_cur + plo);          // both key halves of the row")
        src = src[:i] + "        float plo = psum * (kofs * (1.0f / 64.0f));\n" + src[j:]
    if v in ("noflush", "nothing"):
        src = sub1(src, "      if (f_end <= k_end) {                                   // the current frame ends inside (or at the end of) this tile",
                   "      if (f_end <= k_end && S < 0) {")
    if v in ("noreadback", "nothing"):
        src = sub1(src, "  if (FR == 1 && out_kind == 0) {                             // whole unit", "  if (FR == 1 && out_kind == 0 && S < 0) {                    // whole unit")
    return src


def build():
    os.makedirs(EXP, exist_ok=True)
    src = open(os.path.join(PKG, "csrc", "attention3.hip")).read()
    objs = [os.path.join(PKG, "lib", "obj", f) for f in os.listdir(os.path.join(PKG, "lib", "obj")) if f.endswith(".o") and f != "attention3.o"]
    os.makedirs("/tmp/frabl", exist_ok=True)
    procs = []
    for v in VARIANTS:
        p = f"/tmp/frabl/fr_{v}.hip"
        open(p, "w").write(make(src, v))
        procs.append((v, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(PKG, "csrc"),
                                           "-c", "-o", f"/tmp/frabl/fr_{v}.o", p])))
    for v, pr in procs:
        assert pr.wait() == 0, v
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(EXP, f"libmavlm_fr_{v}.so"),
                        f"/tmp/frabl/fr_{v}.o"] + objs, check=True)
        print("built", v, flush=True)


def run_one():
    sys.path.insert(0, ROOT)
    import torch
    import memory_augmented_vlm_amd  # noqa: F401
    from memory_augmented_vlm_amd import _ops as ops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from diag_vs_hipblaslt_util import timeit_pair
    H, R, S = 8, 12544, 6272
    W = H * 128
    torch.manual_seed(0)
    q = (torch.randn(R, W, device="cuda") * 0.5).bfloat16()
    k = (torch.randn(S, W, device="cuda") * 0.5).bfloat16()
    v = torch.randn(S, W, device="cuda").bfloat16()
    tf, tp = timeit_pair(lambda: ops.attention_frames(q, k, v, H, 196), lambda: ops.attention(q, k, v, H), n=20)       # (the same levelled stream-K schedule)
    print(f"{os.environ.get('FR_V', '?'):10s} R{R} S{S} H{H}: frames {tf * 1e6:7.1f} us | plain {tp * 1e6:7.1f} us | {tf / tp:.4f}x", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "one":
        run_one()
    else:
        for rnd in range(2):
            for v in VARIANTS:
                env = dict(os.environ, MAVLM_LIB=os.path.join(EXP, f"libmavlm_fr_{v}.so"), FR_V=v)
                subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, check=False)
