"""Diagnostic: per-phase cycle shares of the attention kernel (needs lib/libmavlm_stamps.so, MAVLM_LIB env)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops
R, S, H, D = 12544, int(sys.argv[1]) if len(sys.argv) > 1 else 6272, 8, 1024
q = torch.randn(R, D, device="cuda").bfloat16(); kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
for _ in range(3):
    ops.attention(q, kv[:, :D], kv[:, D:], H)
torch.cuda.synchronize()
l = ctypes.CDLL(os.environ["MAVLM_LIB"])
buf = (ctypes.c_ulonglong * (8 * 4096))()
assert l.mavlm_debug_read_stamps(buf, 8 * 4096) == 0
a = np.array(buf[:], dtype=np.float64).reshape(4096, 8)[:784 * 4]
nt = a[:, 5]
per = a[:, :5] / nt[:, None]
names = ["QK", "softmax", "PV", "lds-store", "barrier"]
print("cycles per tile per wave (mean over waves):", {n: round(v) for n, v in zip(names, per.mean(0))}, "total", round(per.sum(1).mean()))
first = per[:2048]; last = per[2048:]
print("first-round waves:", {n: round(v) for n, v in zip(names, first.mean(0))}, "total", round(first.sum(1).mean()))
print("second-round waves:", {n: round(v) for n, v in zip(names, last.mean(0))}, "total", round(last.sum(1).mean()))
wid = a[:, 6].astype(np.int64) & 0xF
print("wave slot histogram:", np.bincount(wid)[:8])
