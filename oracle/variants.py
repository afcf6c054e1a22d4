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


# ---- TemporalGRUEncoder (bigru.py:14-75) ---------------------------------------------------------------------------
def gru_weights(D, H, seed=95, grid="bf16"):
    """nn.GRU(D, H, bidirectional) parameters, PyTorch init scale U(-1/sqrt(H), 1/sqrt(H)), state-dict names."""
    r = O.rounder(grid)
    b = 1.0 / np.sqrt(H)
    w = {}
    n = seed * 1000
    for sfx in ("", "_reverse"):
        for name, shape in (("weight_ih_l0", (3 * H, D)), ("weight_hh_l0", (3 * H, H)), ("bias_ih_l0", (3 * H,)),
                            ("bias_hh_l0", (3 * H,))):
            n += 1
            w["gru." + name + sfx] = r(O.hash_uniform(shape, n, -b, b))
    return w


def sine_time_table(max_frames, dim):
    pos = np.arange(max_frames, dtype=np.float32)[:, None]
    freq = np.exp(np.arange(0, dim, 2, dtype=np.float32) * F32(-np.log(10000.0) / dim)).astype(F32)
    pe = np.zeros((max_frames, dim), F32)
    pe[:, 0::2] = np.sin(pos * freq)
    pe[:, 1::2] = np.cos(pos * freq)
    return pe


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_encoder(x, w, H, mode="fp32", use_pe=False, max_frames=300):
    """x [F,P,D] -> x + broadcast(biGRU(mean_p x))  (bigru.py:45-75).  Emulation modes round where the HIP path stores
    16 bits: the frame means, (PE sum), the GRU outputs, the final sum; gate math in float64 -> float32."""
    r = O.rounder(mode)
    Fn, P, D = x.shape
    v = r(x.astype(np.float64).mean(axis=1).astype(F32))
    if use_pe:
        v = r(v + r(sine_time_table(max_frames, D)[:Fn]))
    out = np.zeros((Fn, 2 * H), F32)
    for d, sfx in enumerate(("", "_reverse")):
        Wi, Wh = w["gru.weight_ih_l0" + sfx].astype(np.float64), w["gru.weight_hh_l0" + sfx].astype(np.float64)
        bi, bh = w["gru.bias_ih_l0" + sfx].astype(np.float64), w["gru.bias_hh_l0" + sfx].astype(np.float64)
        xg = (v.astype(np.float64) @ Wi.T + bi).astype(F32).astype(np.float64)      # fp32 GEMM output
        h = np.zeros(H)
        order = range(Fn) if d == 0 else range(Fn - 1, -1, -1)
        for t in order:
            g = Wh @ h + bh
            rr = _sigmoid(xg[t, :H] + g[:H])
            z = _sigmoid(xg[t, H:2 * H] + g[H:2 * H])
            n = np.tanh(xg[t, 2 * H:] + rr * g[2 * H:])
            h = (1 - z) * n + z * h
            out[t, d * H:(d + 1) * H] = h
    ctx = r(out)
    return r(x + ctx[:, None, :])


# ---- scene segmentation / sampling (segment.py:3-53,252-337): float part only; the integer logic is host code of the
# product and is pinned directly against the reference's outputs (tests/test_host_cpu.py) ---------------------------
def frame_means(x):
    return x.astype(np.float64).mean(axis=1).astype(F32)


def adjacent_cosine(v, eps=1e-2):
    a, b = v[:-1].astype(np.float64), v[1:].astype(np.float64)
    na, nb = np.maximum(np.linalg.norm(a, axis=1), eps), np.maximum(np.linalg.norm(b, axis=1), eps)
    return ((a * b).sum(axis=1) / (na * nb)).astype(F32)


def scene_features(T, P, D, scene_len, seed, noise=0.05):
    """Synthetic [T,P,D] frames: one random prototype per scene of `scene_len` frames + small per-frame noise."""
    ns = -(-T // scene_len)
    proto = O.hash_normal_like((ns, 1, D), seed)
    x = proto[np.arange(T) // scene_len] + noise * O.hash_normal_like((T, P, D), seed + 1)
    return O.bf16_round(x)
