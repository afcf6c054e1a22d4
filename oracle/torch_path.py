"""CPU oracle for the GRADIENTS of the memory path (test infrastructure only - never imported by the product).

The reference has no backward code of its own: it runs the forward of
`llava/model/memory_module/MemoryController.py:47-158` and `llava/model/llava_arch.py:545-554` under torch autograd.
This file restates that forward with plain torch ops on CPU (float64 by default) over the oracle's weight
dictionary (reference state-dict names, `oracle/memory_path.make_weights`) and lets autograd differentiate it - the
"plain fp32/fp64 torch reference of the same op" for a floating-point kernel.  Pinned in
tests/test_oracle_backward.py against gradients the imported reference itself produced
(tests/golden/g8_grads_*.npz, generator tests/golden/make_golden.py g8): parity is pinned, not assumed.
"""
import math
from typing import Dict, List

import numpy as np
import torch

PFX = "recurrent_memory_transformer"


def params_from(w: Dict[str, np.ndarray], dtype=torch.float64) -> Dict[str, torch.Tensor]:
    """Leaf tensors (requires_grad) for every floating-point entry of the oracle weight dict but the PE buffer."""
    out = {}
    for k, v in w.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
        if k != "positional_encoding.frame_embed":
            t.requires_grad_(True)
        out[k] = t
    return out


def _lin(x, p, name):
    return torch.nn.functional.linear(x, p[f"{name}.weight"], p[f"{name}.bias"])          # what nn.Linear calls


def _ln(x, p, name, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), p[f"{name}.weight"], p[f"{name}.bias"], eps)   # nn.LayerNorm


def mha(xq, xkv, p, prefix, heads, eps):
    """Attention.forward, MemoryController.py:47-57 (no mask, no dropout in the restated path)."""
    R, D = xq.shape
    d = D // heads
    q = _lin(xq, p, f"{prefix}.q_proj").view(R, heads, d).permute(1, 0, 2)
    k = _lin(xkv, p, f"{prefix}.k_proj").view(-1, heads, d).permute(1, 0, 2)
    v = _lin(xkv, p, f"{prefix}.v_proj").view(-1, heads, d).permute(1, 0, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1)
    ctx = (a @ v).permute(1, 0, 2).reshape(R, D)
    return _ln(_lin(ctx, p, f"{prefix}.residual.dense") + xq, p, f"{prefix}.residual.layernorm", eps)


def layer(m, seg, p, prefix, heads, eps):
    """TransformerLayer.forward, MemoryController.py:69-72."""
    a = mha(m, seg, p, f"{prefix}.memory_segment_fusion_attention", heads, eps)
    h = torch.relu(_lin(a, p, f"{prefix}.mlp.0"))
    return _ln(_lin(h, p, f"{prefix}.residual.dense") + a, p, f"{prefix}.residual.layernorm", eps)


def run_steps(p, cfg, segs: List[np.ndarray], cache_cap: int = 10) -> List[torch.Tensor]:
    """TransformerProjector.forward over the chunks of one video (MemoryController.py:118-158): returns the final
    memory_cache (graph-carrying tensors [M,P,D], oldest first)."""
    dt = p[f"{PFX}.initial_memory"].dtype
    R, D = cfg.mem_rows, cfg.hidden
    cache: List[torch.Tensor] = []
    for seg in segs:
        x = torch.from_numpy(np.ascontiguousarray(seg)).to(dt).reshape(-1, D)
        if cache:
            m = mha(cache[-1].reshape(R, D), torch.cat(cache, dim=0).reshape(-1, D), p,
                    f"{PFX}.memory_update_attention", cfg.heads, cfg.eps)
        else:
            m = (p[f"{PFX}.initial_memory"] + p[f"{PFX}.memory_pos_embed"]).reshape(R, D)
        for l in range(cfg.depth):
            m = layer(m, x, p, f"{PFX}.layers.{l}", cfg.heads, cfg.eps)
        cache.append(m.reshape(cfg.mem_tokens, cfg.patches, D))
        cache = cache[-cache_cap:]
    return cache


def fuse(p, cache: List[torch.Tensor]) -> torch.Tensor:
    """memory_fuser MLP + token-type row 0 (llava_arch.py:545-553): [n*M*P, D]."""
    mem = torch.cat(cache, dim=0)
    u = torch.nn.functional.gelu(_lin(mem, p, "memory_fuser.0"))
    y = _lin(u, p, "memory_fuser.2") + p["token_type_embedding.weight"][0]
    return y.reshape(-1, y.shape[-1])


def cpu_reference_step_timer(cfg, w: Dict[str, np.ndarray], segs: List[np.ndarray], dtype=torch.float32):
    """bench.py's `cpu_baseline` leg: the same ATen call sequence the reference executes on CPU (F.linear, matmul, the
    score division, softmax, F.layer_norm - MemoryController.py:47-57,69-72,118-158), without autograd, in `dtype`
    (float32, or bfloat16 as the reference runs under `torch_dtype=bfloat16`).  Returns a zero-argument callable that
    runs the recurrent steps once and returns the newest memory."""
    with torch.no_grad():
        p = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in w.items()
             if k.startswith(PFX) and v.dtype.kind == "f"}
    xs = [np.ascontiguousarray(s_, dtype=np.float32) for s_ in segs]

    def run():
        with torch.no_grad():
            return run_steps(p, cfg, xs)[-1]
    return run


def grads(p: Dict[str, torch.Tensor], loss: torch.Tensor) -> Dict[str, np.ndarray]:
    names = [k for k, t in p.items() if t.requires_grad]
    gs = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    return {k: (np.zeros(tuple(p[k].shape), np.float32) if g is None else g.detach().to(torch.float32).numpy())
            for k, g in zip(names, gs)}
