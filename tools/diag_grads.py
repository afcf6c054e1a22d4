"""Diagnostic (not a test): per-parameter gradient error of the HIP backward vs the reference fp32 golden, beside the
reference's own bf16 envelope.  usage: python tools/diag_grads.py d256|d1024"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))   # gpu_util
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import memory_augmented_vlm_amd  # noqa
from oracle import memory_path as O
from conftest import load_golden
from test_gpu_backward_path import _segs, _cotangents, _hip_grads
from test_gpu_path import make_projector

tag = sys.argv[1] if len(sys.argv) > 1 else "d256"
z, meta = load_golden(f"g8_grads_{tag}.npz")
cfg = O.PathConfig(hidden=meta["hidden"], heads=meta["heads"], mem_tokens=meta["mem_tokens"], depth=meta["depth"])
w = O.make_weights(cfg, seed=meta["wseed"])
rm = make_projector(cfg, w, "bf16").train()
segs = _segs(cfg, meta["frames"], meta["segseed0"])
cots = _cotangents(cfg, len(segs), meta["gseed0"], meta["gstd"])
loss, g, _ = _hip_grads(rm, segs, cots, "bf16")
print("loss", loss, "ref", float(z["loss"]))
for name, grad in g.items():
    got = grad.reshape(-1)[::meta["stride"]]
    ref = z["g_" + name + "_sample"]
    print("%-68s err %.3e env %.3e ratio %.2f  |g| %.2e" % (name, O.rel_l2(got, ref), float(z["env_" + name]),
          O.rel_l2(got, ref) / float(z["env_" + name]), np.linalg.norm(ref)))
