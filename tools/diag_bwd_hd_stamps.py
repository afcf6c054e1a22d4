"""Diagnostic: where does a 32-row tile of attn_bwd_hd_kernel spend its cycles?  `build` (build container): a copy of the product
source with s_memtime stamps around the phases of the tile (first-phase MFMAs / exchange write / barrier / exchange read + values
0-7 / second phase / vmcnt(0) / barrier), summed per wave -> lib/exp/libmavlm_bhd_stamps.so.  `run` (GPU box): the shares.
Read the SHARES, not the absolute time (the stamps' fences forbid overlaps the real kernel has).  Evidence only.
usage: python tools/diag_bwd_hd_stamps.py build | run"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "memory-augmented-vlm_amd")
EXP = os.path.join(PKG, "lib", "exp")
NS = 8


def build():
    os.makedirs(EXP, exist_ok=True)
    s = open(os.path.join(PKG, "csrc", "attention_bwd_hd.hip")).read()

    def rep(old, new):
        nonlocal s
        assert old in s, old[:80]
        s = s.replace(old, new, 1)
    rep("namespace {\n\ntemplate <int V>\nstruct WIC", '''__device__ unsigned long long g_bhd_stamps[2][12];
extern "C" int mavlm_exp_bhd_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bhd_stamps), sizeof(unsigned long long) * 24);
  if (e != hipSuccess) return (int)e;
  if (reset) { unsigned long long z[24] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_bhd_stamps), z, sizeof(z)); }
  return (int)e;
}
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

namespace {

template <int V>
struct WIC''')
    rep("  auto tile = [&](auto par, int t) {\n", "  unsigned long long sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q0, q1, L0, L1, R0, R1;\n  STAMP(L0);\n  asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(R0) :: \"memory\");\n  auto tile = [&](auto par, int t) {\n    STAMP(q0);\n")
    rep("      bw_for_each(std::make_integer_sequence<int, KS>{}, sstep);\n", "      bw_for_each(std::make_integer_sequence<int, KS>{}, sstep);\n      STAMP(q1); sa[0] += q1 - q0; q0 = q1;\n")
    rep('        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (no vmcnt wait: the DMAs of tile t+1 stay in flight)\n',
        '        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");\n        STAMP(q1); sa[1] += q1 - q0; q0 = q1;\n')
    rep('        __builtin_amdgcn_s_barrier();\n        asm volatile("" ::: "memory");\n#pragma unroll\n        for (int j = 0; j < 4; ++j) {\n          const f32x4 a =',
        '        __builtin_amdgcn_s_barrier();\n        asm volatile("" ::: "memory");\n        STAMP(q1); sa[2] += q1 - q0; q0 = q1;\n#pragma unroll\n        for (int j = 0; j < 4; ++j) {\n          const f32x4 a =')
    rep("      vpack(WIC<0>{});\n", "      vpack(WIC<0>{});\n      STAMP(q1); sa[3] += q1 - q0; q0 = q1;\n")
    rep("      bw_for_each(std::make_integer_sequence<int, NA>{}, zstep);\n    }\n", "      bw_for_each(std::make_integer_sequence<int, NA>{}, zstep);\n      STAMP(q1); sa[4] += q1 - q0; q0 = q1;\n    }\n")
    rep('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave\'s DMAs of tile t+1 have landed\n',
        '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    STAMP(q1); sa[5] += q1 - q0; q0 = q1;\n')
    rep("    __builtin_amdgcn_s_barrier();\n    asm volatile(\"\" ::: \"memory\");\n  };\n\n  int t = 0;", "    __builtin_amdgcn_s_barrier();\n    asm volatile(\"\" ::: \"memory\");\n    STAMP(q1); sa[6] += q1 - q0; sa[7] += 1;\n  };\n\n  int t = 0;")
    rep("  // ---- epilogue: Out[x][h*HD + 32 (SL db + slab)", '''  STAMP(L1);
  asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(R1) :: "memory");
  if (lane == 0) {
    constexpr int MI = (MODE == 0) ? 0 : 1;
    for (int i = 0; i < 8; ++i) atomicAdd(&g_bhd_stamps[MI][i], sa[i]);
    atomicAdd(&g_bhd_stamps[MI][8], L1 - L0);
    atomicAdd(&g_bhd_stamps[MI][9], R1 - R0);
  }
  // ---- epilogue: Out[x][h*HD + 32 (SL db + slab)''')
    os.makedirs("/tmp/bhdabl", exist_ok=True)
    open("/tmp/bhdabl/bhd_stamps.hip", "w").write(s)
    objs = [os.path.join(PKG, "lib", "obj", f) for f in os.listdir(os.path.join(PKG, "lib", "obj")) if f.endswith(".o") and f != "attention_bwd_hd.o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(PKG, "csrc"), "-c", "-o",
                    "/tmp/bhdabl/bhd_stamps.o", "/tmp/bhdabl/bhd_stamps.hip"], check=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(EXP, "libmavlm_bhd_stamps.so"),
                    "/tmp/bhdabl/bhd_stamps.o"] + objs, check=True)
    print("built", flush=True)


def run():
    sys.path.insert(0, ROOT)
    os.environ["MAVLM_LIB"] = os.path.join(EXP, "libmavlm_bhd_stamps.so")
    import torch
    import memory_augmented_vlm_amd  # noqa: F401
    from memory_augmented_vlm_amd import _capi as capi, _ops as ops
    lib = capi.lib()
    lib.mavlm_exp_bhd_stamps.restype = ctypes.c_int
    lib.mavlm_exp_bhd_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    hd, H, R, S = 448, 8, 1568, 1568 + 32 * 196
    W = H * hd
    torch.manual_seed(0)
    q = (torch.randn(R, W, device="cuda") * 0.5).bfloat16()
    k = (torch.randn(S, W, device="cuda") * 0.5).bfloat16()
    v = torch.randn(S, W, device="cuda").bfloat16()
    do = (torch.randn(R, W, device="cuda") * 0.5).bfloat16()
    o, lse = ops.attention(q, k, v, H, want_lse=True, head_dim=hd)
    scale = ops.attn_scale(hd)
    for _ in range(3):
        ops.attention_bwd_hd(q, k, v, o, do, lse, H, hd, scale)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 24)()
    lib.mavlm_exp_bhd_stamps(buf, 1)
    ops.attention_bwd_hd(q, k, v, o, do, lse, H, hd, scale)
    torch.cuda.synchronize()
    lib.mavlm_exp_bhd_stamps(buf, 0)
    names = ("first phase (28 MFMAs + DMA issue)", "exchange write + lgkmcnt(0)", "barrier 1", "exchange read + values 0-7", "second phase",
             "vmcnt(0)", "barrier 2")
    for mi, kn in enumerate(("dQ (MODE 0)", "dK + dV (MODE 3)")):
        b = buf[12 * mi:12 * mi + 12]
        tiles = b[7]
        tot = sum(b[:7])
        print(f"{kn}: wave-tiles {tiles}")
        for i, n in enumerate(names):
            print(f"  {n:38s} {b[i] / tiles:8.1f} ticks per wave-tile  {100.0 * b[i] / tot:5.1f} %")
        print(f"  {'total':38s} {tot / tiles:8.1f}   clock ratio s_memtime / s_memrealtime x 100 MHz = {b[8] / max(b[9], 1) * 100.0:.0f} MHz")


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
