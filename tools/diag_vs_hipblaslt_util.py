"""Interleaved A/B timing helper of the diagnostics (rule 24)."""
import time

import torch


def timeit_pair(fa, fb, n=30, rounds=5):
    """Both candidates are warmed up for ~0.25 s first (the first kernel measured after an idle gap runs 10-20 % slower while
    the clocks ramp), then `rounds` alternating blocks of n launches each; medians (seconds per launch)."""
    t0 = time.time()
    while time.time() - t0 < 0.25:
        for _ in range(10):
            fa()
        for _ in range(10):
            fb()
        torch.cuda.synchronize()
    ta, tb = [], []
    for _ in range(rounds):
        for f, acc in ((fa, ta), (fb, tb)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            torch.cuda.synchronize()
            acc.append(e0.elapsed_time(e1) / n * 1e-3)
    ta.sort()
    tb.sort()
    return ta[len(ta) // 2], tb[len(tb) // 2]
