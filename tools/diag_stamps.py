"""Diagnostic: where does a 64-key tile of attn_fwd3 spend its cycles?  Needs lib/exp/libmavlm_stamps.so - a build of
the library from a scratch copy of csrc/ in which the tile loop is bracketed with s_memtime stamps (DMA issue / phase A /
phase B / rescale decision / vmcnt wait / barrier), summed per wave and added into a __device__ array by lane 0.  Read the
SHARES, not the absolute time (the stamps' fences forbid overlaps the real kernel has).  Evidence only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MAVLM_LIB"] = os.path.join(ROOT, "memory-augmented-vlm_amd", "lib", "exp", "libmavlm_stamps.so")
import torch
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

lib = capi.lib()
lib.mavlm_exp_stamps.restype = ctypes.c_int
lib.mavlm_exp_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
R, S, H, D = 12544, 6272, 8, 1024
q = torch.randn(R, D, device="cuda").bfloat16()
kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
for _ in range(3):
    ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 12)()
lib.mavlm_exp_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.attention(q, kv[:, :D], kv[:, D:], H, want_lse=True)
e1.record()
torch.cuda.synchronize()
lib.mavlm_exp_stamps(buf, 0)
names = ("dma issue", "phase A (QK^T + exp)", "phase B (PV + sums/max)", "rescale decision", "vmcnt(0) wait", "barrier")
tiles = buf[6]
tot = sum(buf[i] for i in range(6))
print(f"kernel {e0.elapsed_time(e1)*1e3:.1f} us (stamped build); wave-tiles {tiles}")
for i, n in enumerate(names):
    print(f"{n:28s} {buf[i]/tiles:8.1f} cycles per wave-tile  {100.0*buf[i]/tot:5.1f} %")
print(f"{'total':28s} {tot/tiles:8.1f}")
waves = tiles / ((S + 63) // 64)
if buf[9]:
    print(f"in-kernel clock over the tile loop: {buf[8] / buf[9] * 100.0:.0f} MHz (s_memtime / s_memrealtime x 100 MHz, mean over waves); "
          f"loop wall time per wave {buf[9] / waves / 100.0:.1f} us")
print(f"calibration: 32 x s_nop 15 (512 wait states) read {buf[7]/waves:.1f} ticks per wave (incl. one stamp's own cost)")
