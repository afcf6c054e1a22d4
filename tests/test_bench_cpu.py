"""CPU: bench.py's launcher refuses to run fewer ranks than asked (no GPU in this container: every N>1 request must exit
non-zero with a message), and the timed CPU-baseline code path equals the validated numpy oracle."""
import os
import subprocess
import sys

import numpy as np
import torch

from conftest import ROOT


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=300)


def test_gpus_2_without_enough_devices_fails_loudly():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout                     # no JSON line pretending to be a result


def test_launcher_and_flag_disagree_fails_loudly():
    r = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_rehearsal_needs_a_gpu_too():
    r = _run(["--gpus", "2", "--rehearse-gloo"])
    assert r.returncode != 0


def test_cpu_baseline_code_path_equals_the_oracle():
    """bench.py times oracle/torch_path.cpu_reference_step_timer; same numbers as the numpy oracle (fp32 mode), which is
    itself pinned to the reference's outputs (tests/test_oracle_golden.py)."""
    from oracle import memory_path as O
    from oracle import torch_path as TP
    cfg = O.PathConfig(hidden=128, heads=2, mem_tokens=2, depth=2)
    w = O.make_weights(cfg, seed=5)
    segs = [O.bf16_round(O.hash_normal_like((2, 196, 128), 50 + t)) for t in range(2)]
    rm = O.RecurrentMemory(cfg, w, "fp32")
    rm.reset()
    for s_ in segs:
        cache, _ = rm.step(s_)
    got = TP.cpu_reference_step_timer(cfg, w, segs, torch.float32)().numpy()
    assert O.rel_l2(got, cache[-1]) < 1e-5
    got16 = TP.cpu_reference_step_timer(cfg, w, segs, torch.bfloat16)().float().numpy()
    assert O.rel_l2(got16, cache[-1]) < 3e-2             # the reference's own bf16 envelope
