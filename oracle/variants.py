"""CPU oracle for the INACTIVE variants of the reference - test infrastructure only.

  transformer_fuser   `MemoryFuser` (llava/model/memory_module/MemoryFuser.py:4-30): Linear -> nn.TransformerEncoder
                      (post-norm layers, nhead 4, ff 4D, exact GELU, LayerNorm eps 1e-5) -> Linear.
Pinned in tests/test_oracle_golden.py against outputs of the imported reference class (tests/golden/g9_variants.npz).
"""
import numpy as np

from . import memory_path as O

F32 = np.float32


def transformer_fuser(x, w, heads=4, mode="fp32", layers=2, eps=1e-5, return_stages=False):
    """x [N, D]; w: state-dict of the reference MemoryFuser (numpy).  Emulation modes round where the HIP path stores
    16-bit tensors: every GEMM output, the attention context, the LayerNorm outputs."""
    r = O.rounder(mode)
    D = x.shape[1]
    y = r(O.linear(r(x), w["input_proj.weight"], w["input_proj.bias"]))
    stages = [y]
    for l in range(layers):
        p = f"transformer_encoder.layers.{l}."
        qkv = r(O.linear(y, w[p + "self_attn.in_proj_weight"], w[p + "self_attn.in_proj_bias"]))
        ctx, _, _, _ = O.attention_heads(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], heads, mode,
                                         kv_tile=32 if mode != "fp32" else None, wave_rows=16 if mode != "fp32" else None)
        ctx = r(ctx)
        pre = O.linear(ctx, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"]) + y
        y = r(O.layernorm(pre, w[p + "norm1.weight"], w[p + "norm1.bias"], eps))
        h = r(O.gelu_erf(O.linear(y, w[p + "linear1.weight"], w[p + "linear1.bias"])))
        pre = O.linear(h, w[p + "linear2.weight"], w[p + "linear2.bias"]) + y
        y = r(O.layernorm(pre, w[p + "norm2.weight"], w[p + "norm2.bias"], eps))
        stages.append(y)
    out = r(O.linear(y, w["output_proj.weight"], w["output_proj.bias"]))
    return (out, stages) if return_stages else out


def fuser_weights(D, layers=2, seed=91, grid="bf16"):
    """Synthetic weights with the reference MemoryFuser's state-dict names and PyTorch's init scales."""
    r = O.rounder(grid)
    w = {}
    ctr = [seed * 1000]

    def nxt():
        ctr[0] += 1
        return ctr[0]

    def lin(name, o, i, wname="weight", bname="bias"):
        b = 1.0 / np.sqrt(i)
        w[f"{name}.{wname}" if wname == "weight" else f"{name}{wname}"] = r(O.hash_uniform((o, i), nxt(), -b, b))
        w[f"{name}.{bname}" if bname == "bias" else f"{name}{bname}"] = r(O.hash_uniform((o,), nxt(), -b, b))

    lin("input_proj", D, D)
    lin("output_proj", D, D)
    for l in range(layers):
        p = f"transformer_encoder.layers.{l}"
        lin(p + ".self_attn.in_proj", 3 * D, D, "_weight", "_bias")
        lin(p + ".self_attn.out_proj", D, D)
        lin(p + ".linear1", 4 * D, D)
        lin(p + ".linear2", D, 4 * D)
        for n in ("norm1", "norm2"):
            w[f"{p}.{n}.weight"] = r(1.0 + O.hash_uniform((D,), nxt(), -0.1, 0.1))
            w[f"{p}.{n}.bias"] = r(O.hash_uniform((D,), nxt(), -0.1, 0.1))
    return w
