#!/bin/bash
# builds lib/exp/libmavlm_stamps.so from the CURRENT csrc with s_memtime stamps in the attn_fwd3 tile loop
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
[ -d "$ROOT/memory-augmented-vlm_amd/csrc" ] || { echo "repository root not found: $ROOT" >&2; exit 1; }
rm -rf /tmp/stamp && mkdir -p /tmp/stamp && cp -r "$ROOT/memory-augmented-vlm_amd/csrc" /tmp/stamp/csrc && mkdir -p /tmp/stamp/include && cp "$ROOT/include/mavlm.h" /tmp/stamp/include/
cd /tmp/stamp/csrc && sed -i 's#"../../include/mavlm.h"#"../include/mavlm.h"#' *.hip *.h
python - <<'EOF'
p='/tmp/stamp/csrc/attention3.hip'
s=open(p).read()
def rep(old,new):
    global s
    assert old in s, old[:80]
    s=s.replace(old,new,1)
rep('''namespace {

constexpr int HD3 = 128, KT3 = 64;''','''__device__ unsigned long long g_stamps[12];
extern "C" int mavlm_exp_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 12);
  if (e != hipSuccess) return (int)e;
  if (reset) { unsigned long long z[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
  return (int)e;
}
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

namespace {

constexpr int HD3 = 128, KT3 = 64;''')
rep('''  float m_run = -1e30f, l_run = 0.f;
''','''  float m_run = -1e30f, l_run = 0.f;
  unsigned long long acc_dma = 0, acc_a = 0, acc_b = 0, acc_dec = 0, acc_vm = 0, acc_bar = 0, t0, t1, t2, t3, t4, t5, t6, c0, c1;
  STAMP(c0);
  asm volatile("s_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15\\n\\ts_nop 15" ::: "memory");
  STAMP(c1);
  unsigned long long L0, L1, R0, R1;
  STAMP(L0);
  asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(R0) :: "memory");
''')
rep('''    const bool has_next = t + 1 < nt;
''','''    const bool has_next = t + 1 < nt;
    STAMP(t0);
''')
rep('''    const int ktile = (t + 2) * KT3 * ldk * 2, vtile = (t + 1) * KT3 * ldv * 2;     // scalar byte offsets of K(t+2), V(t+1)
''','''    const int ktile = (t + 2) * KT3 * ldk * 2, vtile = (t + 1) * KT3 * ldv * 2;     // scalar byte offsets of K(t+2), V(t+1)
    STAMP(t1);
''')
rep('''    // [B] O^T += V(t)^T.P(t)^T''','''    STAMP(t2);
    // [B] O^T += V(t)^T.P(t)^T''')
rep('''    l_run += psum;
''','''    l_run += psum;
    STAMP(t3);
''')
rep('''    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMAs of K(t+2), V(t+1) have landed
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };''','''    STAMP(t4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMAs of K(t+2), V(t+1) have landed
    STAMP(t5);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP(t6);
    acc_dma += t1 - t0; acc_a += t2 - t1; acc_b += t3 - t2; acc_dec += t4 - t3; acc_vm += t5 - t4; acc_bar += t6 - t5;
  };''')
rep('''  // ---- epilogue: O[q][h*128 + 32db + 8g + 4hh + 0..3] = O^T / l''','''  STAMP(L1);
  asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(R1) :: "memory");
  if (lane == 0) {
    atomicAdd(&g_stamps[8], L1 - L0); atomicAdd(&g_stamps[9], R1 - R0);
    atomicAdd(&g_stamps[0], acc_dma); atomicAdd(&g_stamps[1], acc_a); atomicAdd(&g_stamps[2], acc_b);
    atomicAdd(&g_stamps[3], acc_dec); atomicAdd(&g_stamps[4], acc_vm); atomicAdd(&g_stamps[5], acc_bar);
    atomicAdd(&g_stamps[6], (unsigned long long)nt); atomicAdd(&g_stamps[7], c1 - c0);
  }
  // ---- epilogue: O[q][h*128 + 32db + 8g + 4hh + 0..3] = O^T / l''')
open(p,'w').write(s)
EOF
mkdir -p "$ROOT/memory-augmented-vlm_amd/lib/exp"
# (a failed build fails the script: no pipe that swallows hipcc's status)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o "$ROOT/memory-augmented-vlm_amd/lib/exp/libmavlm_stamps.so" gemm.hip gemm256.hip gemm256p.hip attention.hip attention3.hip attention_hd.hip attention_bwd.hip backward.hip variants.hip elementwise.hip mavlm_api.hip prof.hip
ls -la "$ROOT/memory-augmented-vlm_amd/lib/exp/libmavlm_stamps.so"
