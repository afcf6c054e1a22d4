"""Diagnostic: race screen of the 256-tile GEMM schedules (cdna_hip_programming.md: "a sync-structure edit makes a NEW
template: screen it for races over many runs at several sizes").  Integer-valued operands (every product and partial sum is
exact in fp32), many launches per shape, both tile kernels and both tile heights, compared bit for bit with a torch
integer reference.  usage: python tools/diag_gemm_race_screen.py [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memory_augmented_vlm_amd
from memory_augmented_vlm_amd import _capi as capi, _ops as ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = capi.lib()
torch.manual_seed(0)
bad = 0
for (M, N, K) in [(12544, 1024, 1024), (6272, 4096, 1024), (12544, 1024, 4096), (1000, 512, 192), (3136, 2048, 2048), (777, 256, 64),
                  (25088, 1024, 128), (1568, 3584, 14336)]:
    a = torch.randint(-3, 4, (M, K), device="cuda").to(torch.bfloat16)
    w = torch.randint(-3, 4, (N, K), device="cuda").to(torch.bfloat16)
    b = torch.randint(-8, 9, (N,), device="cuda").float()
    ref = (a.float() @ w.float().t() + b)          # |values| <= 9 * K + 8 < 2^24: exact in fp32
    for tile in (256, 257, 129):                   # 129: the 128x256 two-per-CU kernel (round 4; the row hook does not apply)
        for rows in ((224, 256) if tile != 129 else (0,)):
            capi.check(lib.mavlm_set_gemm_tile(tile), "tile"); capi.check(lib.mavlm_set_gemm_rows(rows), "rows")
            out = torch.empty(M, N, device="cuda", dtype=torch.float32)
            n_bad = 0
            for _ in range(reps):
                out.fill_(float("nan"))
                ops.linear(a, w, b, capi.EPI_F32, out=out)
                if not torch.equal(out, ref):
                    n_bad += 1
            bad += n_bad
            print(f"M{M} N{N} K{K} tile {tile} rows {rows}: {reps - n_bad}/{reps} exact", flush=True)
lib.mavlm_set_gemm_tile(0); lib.mavlm_set_gemm_rows(0)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} wrong results)")
sys.exit(1 if bad else 0)
