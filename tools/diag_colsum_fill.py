"""Diagnostic: column-sum pass (R = 12544, H = 8) at several key counts and workgroup counts of its balanced schedule
(mavlm_set_attention_colsum_wgs).  History: with whole / half units per workgroup S = 6144 (768 workgroups, 3 per CU) ran at
904 TFLOP/s, the bench shape S = 6272 (784: 16 CUs get a 4th) at 820, S = 8192 (512 whole units, 2 per CU) at 975."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

R, H, D = 12544, 8, 1024
torch.manual_seed(0)
q = torch.randn(R, D, device="cuda").bfloat16()
cases = [(6272, g) for g in (0, 0, 256, 384, 512, 640, 768, 1024)] + [(6144, 0), (8192, 0), (4096, 0), (1568, 0)]
if os.environ.get("COLSUM_QUICK"):
    cases = [(6272, 0), (6272, 0), (8192, 0), (6272, 0), (8192, 0)]       # (the first line of a process is a warm-up)
for S, G in cases:
    capi.check(capi.lib().mavlm_set_attention_colsum_wgs(G), "set")
    kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
    _, lse = ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True)
    for _ in range(5):
        ops.attention_colsum(q, kv[:, :D], lse, H)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attention_colsum(q, kv[:, :D], lse, H)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    ms = sorted(ts)[2]
    print(f"S={S:6d} wgs={G:5d} {ms * 1e3:8.1f} us {2.0 * R * S * D / ms / 1e9:7.1f} TF", flush=True)
capi.lib().mavlm_set_attention_colsum_wgs(0)
