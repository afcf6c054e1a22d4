"""Diagnostic: race / determinism screen of the kernels whose synchronisation changed in round 2 (LDS-DMA staging with
hand-counted waits, levelled stream-K + merge, balanced column sums, the pipelined attention backward).  All of them are
deterministic by construction (static schedules, no atomics): every launch of a shape must reproduce the first one bit for
bit - a race shows up as a mismatch.  Parity with the oracle is the job of tests/; this only repeats launches.
usage: python tools/diag_determinism_screen.py [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
bad = 0


def screen(name, fn):
    global bad
    first = [t.clone() for t in fn()]
    n_bad = 0
    for _ in range(reps):
        cur = fn()
        if not all(torch.equal(a, b) for a, b in zip(first, cur)):
            n_bad += 1
    bad += n_bad
    print(f"{name:58s} {reps - n_bad}/{reps} identical", flush=True)


H, D = 8, 1024
for (R, S) in [(12544, 6272), (12544, 12544), (1568, 6272), (4100, 64 * 9 + 5)]:
    q = torch.randn(R, D, device="cuda").bfloat16()
    kv = torch.randn(S, 2 * D, device="cuda").bfloat16()
    k, v = kv[:, :D], kv[:, D:]
    screen(f"attention fwd R={R} S={S}", lambda: ops.attention(q, k, v, H, want_lse=True))
    o, lse = ops.attention(q, k, v, H, want_lse=True)
    screen(f"column sums R={R} S={S}", lambda: (ops.attention_colsum(q, k, lse, H),))
    do = torch.randn(R, D, device="cuda").bfloat16()
    screen(f"attention bwd R={R} S={S}", lambda: ops.attention_bwd(q, k, v, o, do, lse, H))
Hh, hd = 8, 448
for (R, S) in [(1568, 6272), (3136, 1000)]:
    q = torch.randn(R, Hh * hd, device="cuda").bfloat16()
    kv = torch.randn(S, 2 * Hh * hd, device="cuda").bfloat16()
    screen(f"wide-head attention fwd R={R} S={S}", lambda: ops.attention(q, kv[:, :Hh * hd], kv[:, Hh * hd:], Hh, want_lse=True, head_dim=hd))
print("DETERMINISM SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} mismatching launches)")
sys.exit(1 if bad else 0)
