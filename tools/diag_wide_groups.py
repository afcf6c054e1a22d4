"""Diagnostic (GPU box): the head_dim-448 attention forward, 16-query waves (mavlm_set_attention_wide_groups(1)) against the
pipelined 32-query-wave kernel (2; 3 / 4 = deeper fragment read-ahead), interleaved blocks, each against a torch fp32 reference."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memory_augmented_vlm_amd import _capi as capi  # noqa: E402
from memory_augmented_vlm_amd import _ops as ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = capi.lib()
Hh, hd = 8, 448
MODES = [int(v) for v in os.environ.get("MODES", "1,2").split(",")]
SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ.get(
    "SHAPES", "64x6272,1568x6272,1568x18816,12544x6272,12544x18816,1000x777,100x31,129x33").split(",")]
for (Rr, Ss) in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(Rr + Ss)
    q = torch.randn(Rr, Hh * hd, generator=g).to(dev).bfloat16()
    kv = torch.randn(Ss, 2 * Hh * hd, generator=g).to(dev).bfloat16()
    ref = None
    if Rr * Ss <= 1568 * 6272:
        qf = q.float().view(Rr, Hh, hd).transpose(0, 1)
        kf = kv[:, :Hh * hd].float().view(Ss, Hh, hd).transpose(0, 1)
        vf = kv[:, Hh * hd:].float().view(Ss, Hh, hd).transpose(0, 1)
        sc = (qf @ kf.transpose(1, 2)) / hd ** 0.5
        ref = (torch.softmax(sc, dim=-1) @ vf).transpose(0, 1).reshape(Rr, Hh * hd)
        ref_lse = torch.logsumexp(sc, dim=-1) * 1.4426950408889634
    outs = {}
    res = {m: [] for m in MODES}
    for rnd in range(3):
        for m in MODES:
            lib.mavlm_set_attention_wide_groups(m)
            o, lse = ops.attention(q, kv[:, :Hh * hd], kv[:, Hh * hd:], Hh, want_lse=True, head_dim=hd)
            outs[m] = (o.clone(), lse.clone())
            torch.cuda.synchronize()
            n = 10 if Rr * Ss < 5e7 else 3
            t0 = time.perf_counter()
            for _ in range(n):
                ops.attention(q, kv[:, :Hh * hd], kv[:, Hh * hd:], Hh, want_lse=True, head_dim=hd)
            torch.cuda.synchronize()
            res[m].append((time.perf_counter() - t0) / n)
    fl = 4.0 * Rr * Ss * Hh * hd
    line = f"R={Rr} S={Ss}:"
    for m in MODES:
        line += f"  [{m}] {fl / min(res[m]) / 1e12:7.1f} TF"
        if ref is not None:
            line += f" err {(outs[m][0].float() - ref).abs().max().item():.2e}/{(outs[m][1] - ref_lse).abs().max().item():.1e}"
        elif m != MODES[0]:
            line += f" d {(outs[m][0].float() - outs[MODES[0]][0].float()).abs().max().item():.2e}"
        if torch.isnan(outs[m][0].float()).any():
            line += " NaN!"
    print(line, flush=True)
lib.mavlm_set_attention_wide_groups(0)
