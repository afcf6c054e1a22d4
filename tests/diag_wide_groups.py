"""Diagnostic (GPU box): the head_dim-448 attention forward with one / two query groups per wave (mavlm_set_attention_wide_groups),
interleaved blocks, and that the two forms agree bit for bit."""
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
for (Rr, Ss) in ((64, 6272), (1568, 6272), (1568, 18816), (12544, 6272), (12544, 18816), (1000, 777)):
    g = torch.Generator(device="cpu").manual_seed(Rr + Ss)
    q = torch.randn(Rr, Hh * hd, generator=g).to(dev).bfloat16()
    kv = torch.randn(Ss, 2 * Hh * hd, generator=g).to(dev).bfloat16()
    outs = {}
    res = {1: [], 2: []}
    for rnd in range(4):
        for qg in (1, 2):
            lib.mavlm_set_attention_wide_groups(qg)
            o, lse = ops.attention(q, kv[:, :Hh * hd], kv[:, Hh * hd:], Hh, want_lse=True, head_dim=hd)
            outs[qg] = (o.clone(), lse.clone())
            torch.cuda.synchronize()
            n = 10 if Rr * Ss < 5e7 else 3
            t0 = time.perf_counter()
            for _ in range(n):
                ops.attention(q, kv[:, :Hh * hd], kv[:, Hh * hd:], Hh, want_lse=True, head_dim=hd)
            torch.cuda.synchronize()
            res[qg].append((time.perf_counter() - t0) / n)
    same = torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])
    fl = 4.0 * Rr * Ss * Hh * hd
    print(f"R={Rr} S={Ss}: 16-query waves {fl / min(res[1]) / 1e12:7.1f} TF   32-query waves {fl / min(res[2]) / 1e12:7.1f} TF   bit-identical {same}",
          flush=True)
lib.mavlm_set_attention_wide_groups(0)
