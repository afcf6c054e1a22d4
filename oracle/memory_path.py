"""CPU oracle for the recurrent memory-token + Memory-Fuser path.

TEST INFRASTRUCTURE ONLY.  This file is a plain numpy restatement of the
reference algorithm (reference = /root/reference, 1023604540/Memory-Augmented-VLM).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported baseline.  The
product path (``memory-augmented-vlm_amd/``) never imports anything from
``oracle/`` and has no CPU fallback.

Parity pin: the reference ships no tests and no golden vectors, so
this restatement is pinned against outputs of the reference itself, generated in
the build container by ``tests/golden/make_golden.py`` (imports the reference's
own Python on CPU) and committed under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks fp32 mode against them to <= 1e-5 rel-L2.

Two arithmetic modes:

* ``fp32``  - every op in float32 (matmuls accumulate in float32 via BLAS).
* ``bf16`` / ``fp16`` - *operand-rounding emulation* of the HIP path: float32
  accumulation everywhere, values rounded to the bf16/fp16 grid exactly where the
  HIP kernels round (MFMA operands = every tensor that is stored to HBM between
  kernels, plus the softmax probabilities fed to the P*V MFMA).  See DESIGN.md
  "Rounding points".

Each function cites the reference file:line it restates (paths relative to
/root/reference/).
"""
from __future__ import annotations

import math
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

try:  # scipy ships in the image; keep a pure-numpy erf as a guard
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float32])

F32 = np.float32

# Accumulation width of every contraction.  float32 is the restatement proper; float64 is used by the tests to
# measure the *noise floor* of a 16-bit chain: two correct implementations that differ only in summation order
# decorrelate at every storage rounding (a perturbation d << ulp becomes ~sqrt(d*ulp) after rounding).
_ACC = [np.float32]


class accumulate_in:
    """Context manager: run the oracle with float64 (or float32) accumulation in all matmuls."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = _ACC[0]
        _ACC[0] = self.dtype

    def __exit__(self, *a):
        _ACC[0] = self.prev


def _mm(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    acc = _ACC[0]
    if acc is np.float32:
        return (a.astype(F32) @ b.astype(F32)).astype(F32)
    return (a.astype(acc) @ b.astype(acc)).astype(F32)


# --------------------------------------------------------------------------- #
# rounding helpers
# --------------------------------------------------------------------------- #
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest bfloat16 (ties-to-even); returns float32."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32)
    rounded = ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))))
               & np.uint32(0xFFFF0000))
    out = rounded.view(F32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = np.nan
    return out


def fp16_round(x: np.ndarray) -> np.ndarray:
    return np.asarray(x, dtype=F32).astype(np.float16).astype(F32)


def rounder(mode: str):
    if mode == "fp32":
        return lambda a: np.asarray(a, dtype=F32)
    if mode == "bf16":
        return bf16_round
    if mode == "fp16":
        return fp16_round
    raise ValueError(f"unknown mode {mode!r}")


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 (already on the bf16 grid or not) -> uint16 bf16 bit pattern (RNE)."""
    return (bf16_round(x).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def bits_to_f32(bits: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit pattern -> float32."""
    return (bits.astype(np.uint32) << np.uint32(16)).view(F32)


# --------------------------------------------------------------------------- #
# deterministic counter-based input generator (integer hash -> float)
# --------------------------------------------------------------------------- #
def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32 integer hash on uint32 arrays (wrap-around arithmetic)."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def hash_uniform(shape: Sequence[int], seed: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """U[lo,hi) float32 tensor; element i = f(hash(i, seed)).  Reproducible anywhere."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint32)
        h = _mix32(idx ^ _mix32(np.full(1, seed & 0xFFFFFFFF, dtype=np.uint32) + np.uint32(0x9E3779B9)))
    u = (h >> np.uint32(8)).astype(F32) * F32(1.0 / (1 << 24))  # [0,1), 24 bits: exact in f32
    return (F32(lo) + u * F32(hi - lo)).reshape(shape).astype(F32)


def hash_normal_like(shape: Sequence[int], seed: int, std: float = 1.0) -> np.ndarray:
    """Unit-variance, N(0,1)-like tensor: U(-sqrt3, sqrt3)."""
    s = math.sqrt(3.0) * std
    return hash_uniform(shape, seed, -s, s)


# --------------------------------------------------------------------------- #
# configuration and weights
# --------------------------------------------------------------------------- #
@dataclass
class PathConfig:
    """Hyper-parameters.  Defaults = the values hard-coded at llava/model/llava_arch.py:117-129
    (H=8, P=196, M=8, depth=2, I=4D, eps=1e-12) and :146 (max_frames=600)."""
    hidden: int = 1024
    heads: int = 8
    patches: int = 196
    mem_tokens: int = 8
    depth: int = 2
    inter: Optional[int] = None
    eps: float = 1e-12
    max_frames: int = 600
    cache_cap: int = 10
    chunk: int = 32
    fine_frames: int = 32

    def __post_init__(self):
        if self.inter is None:
            self.inter = 4 * self.hidden
        assert self.hidden % self.heads == 0

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def mem_rows(self) -> int:
        return self.mem_tokens * self.patches


# Qwen2 tokenizer ids of the two hard-coded prompts, llava/model/llava_arch.py:708,714
MEM_PROMPT_IDS = [1986, 374, 264, 1550, 11591, 12126, 315, 279, 2766, 25]
FRAME_PROMPT_IDS = [9485, 525, 48876, 9124, 14087, 504, 279, 2766, 25]


def attn_keys(prefix: str) -> List[str]:
    return [f"{prefix}.{p}.{w}" for p in ("q_proj", "k_proj", "v_proj", "residual.dense", "residual.layernorm")
            for w in ("weight", "bias")]


def make_weights(cfg: PathConfig, seed: int = 1234, grid: str = "bf16") -> Dict[str, np.ndarray]:
    """Synthetic weights with the reference's state-dict names and the init scales of
    SURVEY.md §8d: nn.Linear default U(-1/sqrt(fan_in), +), LN gamma=1±0.1, beta=±0.1, xavier
    initial_memory, N(0,1) pos-embed, N(0,0.02) type embedding.  Values snapped to ``grid``."""
    r = rounder(grid) if grid != "fp32" else (lambda a: np.asarray(a, dtype=F32))
    D, I, M, P = cfg.hidden, cfg.inter, cfg.mem_tokens, cfg.patches
    w: Dict[str, np.ndarray] = {}
    ctr = [seed * 1000]

    def nxt():
        ctr[0] += 1
        return ctr[0]

    def lin(name, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        w[f"{name}.weight"] = r(hash_uniform((out_f, in_f), nxt(), -b, b))
        w[f"{name}.bias"] = r(hash_uniform((out_f,), nxt(), -b, b))

    def ln(name):
        w[f"{name}.weight"] = r(1.0 + hash_uniform((D,), nxt(), -0.1, 0.1))
        w[f"{name}.bias"] = r(hash_uniform((D,), nxt(), -0.1, 0.1))

    def attn(prefix):
        for p in ("q_proj", "k_proj", "v_proj"):
            lin(f"{prefix}.{p}", D, D)
        lin(f"{prefix}.residual.dense", D, D)
        ln(f"{prefix}.residual.layernorm")

    T = "recurrent_memory_transformer"
    xb = math.sqrt(6.0 / (P * D + M * D))  # xavier_uniform on [M,P,D]: fan_in=P*D, fan_out=M*D
    w[f"{T}.initial_memory"] = r(hash_uniform((M, P, D), nxt(), -xb, xb))
    w[f"{T}.memory_pos_embed"] = r(hash_normal_like((M, 1, D), nxt()))
    for l in range(cfg.depth):
        attn(f"{T}.layers.{l}.memory_segment_fusion_attention")
        lin(f"{T}.layers.{l}.mlp.0", I, D)
        lin(f"{T}.layers.{l}.residual.dense", D, I)
        ln(f"{T}.layers.{l}.residual.layernorm")
    attn(f"{T}.memory_update_attention")
    lin("memory_fuser.0", I, D)
    lin("memory_fuser.2", D, I)
    w["token_type_embedding.weight"] = r(hash_normal_like((2, D), nxt(), 0.02))
    w["positional_encoding.frame_embed"] = pe_table(cfg.max_frames, D)
    w["image_newline"] = r(hash_normal_like((D,), nxt(), 0.02))
    return w


# --------------------------------------------------------------------------- #
# a1  temporal positional encoding
# --------------------------------------------------------------------------- #
def pe_table(max_frames: int, dim: int) -> np.ndarray:
    """llava/model/memory_module/position_encoding.py:29-35 (sinusoidal branch), float32 math."""
    pos = np.arange(max_frames, dtype=F32)[:, None]
    div = np.exp(np.arange(0, dim, 2, dtype=F32) * F32(-(math.log(10000.0) / dim))).astype(F32)
    pe = np.zeros((max_frames, dim), dtype=F32)
    ang = (pos * div).astype(F32)
    pe[:, 0::2] = np.sin(ang)
    pe[:, 1::2] = np.cos(ang)
    return pe


def pe_add(x: np.ndarray, idx: np.ndarray, table: np.ndarray, mode: str = "fp32") -> np.ndarray:
    """position_encoding.py:38-69,71-80: x[T,P,D] + table[idx][:,None,:]; ValueError on range."""
    if x.ndim != 3:
        raise ValueError(f"Expected 3D input, got {x.ndim}D.")
    idx = np.asarray(idx, dtype=np.int64)
    if np.any(idx >= table.shape[0]):
        raise ValueError(f"indices exceed max_frames: max {int(idx.max())} vs limit {table.shape[0]}")
    if np.any(idx < 0):
        raise ValueError(f"indices contains negative values: min {int(idx.min())}")
    r = rounder(mode)
    pe = r(table[idx])  # .to(x.dtype), position_encoding.py:58
    return r(r(x) + pe[:, None, :])


# --------------------------------------------------------------------------- #
# a2 / a3  host-side index math
# --------------------------------------------------------------------------- #
def torch_linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """float32 ``torch.linspace`` as ATen computes it (symmetric halves,
    aten/src/ATen/native/RangeFactories: i < steps/2 ? start+step*i : end-step*(steps-1-i))."""
    if steps == 1:
        return np.array([start], dtype=F32)
    s, e = F32(start), F32(end)
    step = F32((e - s) / F32(steps - 1))
    i = np.arange(steps)
    lo = (s + step * i.astype(F32)).astype(F32)
    hi = (e - step * (steps - 1 - i).astype(F32)).astype(F32)
    return np.where(i < steps // 2, lo, hi).astype(F32)


def subsample_count(num_frames: int) -> int:
    """llava/model/llava_arch.py:437-445: F0<32 -> F0 ; else max(64, (F0//32)*32)."""
    if num_frames < 32:
        return num_frames
    return max(64, (num_frames // 32) * 32)


def subsample_indices(num_frames: int) -> np.ndarray:
    """llava_arch.py:451: linspace(0, F0-1, F).long() (truncation)."""
    return torch_linspace_f32(0, num_frames - 1, subsample_count(num_frames)).astype(np.int64)


def fine_frame_indices(num_frames: int, want: int = 32) -> np.ndarray:
    """llava_arch.py:513-522: round(linspace(0,T-1,min(32,T))) clamped (round-half-even)."""
    n = min(want, num_frames)
    v = np.rint(torch_linspace_f32(0, num_frames - 1, n)).astype(np.int64)
    return np.clip(v, 0, num_frames - 1)


def uniform_segment_variant(T: int, d: int = 32) -> List[int]:
    """llava/model/memory_module/segment.py:169-192."""
    b = [0]
    cur = 0
    while cur + d <= T:
        cur += d
        b.append(cur)
    if cur < T:
        b.append(T)
    return b


# --------------------------------------------------------------------------- #
# a5-a7  building blocks
# --------------------------------------------------------------------------- #
def linear(x: np.ndarray, W: np.ndarray, b: np.ndarray) -> np.ndarray:
    """nn.Linear: x W^T + b, float32 accumulate (operands already on their grid)."""
    return (_mm(x, W.T) + b.astype(F32)).astype(F32)


def layernorm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    """nn.LayerNorm over last dim, biased variance, rsqrt(var+eps) (MemoryController.py:24,28)."""
    x64 = x.astype(np.float64)
    mu = x64.mean(-1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(-1, keepdims=True)
    y = (x64 - mu) / np.sqrt(var + eps)
    return (y * g.astype(np.float64) + b.astype(np.float64)).astype(F32)


def gelu_erf(x: np.ndarray) -> np.ndarray:
    """nn.GELU() default = exact erf form (llava_arch.py:134)."""
    x = x.astype(F32)
    return (0.5 * x * (1.0 + _erf(x / F32(math.sqrt(2.0))))).astype(F32)


KV_TILE = 64          # keys per tile of the HIP flash-attention kernel (csrc/attention.hip)
RESCALE_LOG2 = 8.0    # deferred-rescale threshold of the kernel, log2 units (attention.hip RESCALE_LOG2)
WAVE_ROWS = 32        # queries per wave: the rescale decision is wave-uniform (any lane over threshold)


def kernel_tiling(head_dim: int) -> Tuple[int, int]:
    """(keys per tile, queries per wave) of the HIP attention kernel that serves this head_dim:
    attention3.hip (<= 128, zero-padded to 128): 64 / 32;  attention_hd.hip: 32 / 32 at 448 (attn_fwd_hd2_kernel, round 3),
    32 / 16 at the other wide widths (224, 256: attn_fwd_hd_kernel)."""
    if head_dim <= 128:
        return (KV_TILE, WAVE_ROWS)
    return (32, 32) if (head_dim == 448 and WIDE_GROUPS != 1) else (32, 16)


def split_plan(R: int, S: int, heads: int) -> Tuple[int, int]:
    """(number of key splits, tiles per split) of the head_dim<=128 attention kernel - mirrors
    mavlm_attention_splits (csrc/attention3.hip): grids of fewer than 320 workgroups split the keys."""
    items = -(-R // 128) * heads
    nt = -(-S // KV_TILE)
    ns = 1
    if items < 320 and nt >= 16:
        ns = min(8, 512 // items, nt // 8)
        if ns < 2:
            ns = 1
    tps = -(-nt // ns)
    ns = -(-nt // tps)
    return ns, (tps if ns > 1 else 0)


STREAMK_WGS = 512     # persistent workgroups of the stream-K schedule (csrc/attention3.hip ATTN3_SK_WGS = 256 CUs x 2)
STREAMK_MIN_TILES = 64   # ... used from this many 64-key tiles per unit (mavlm_set_attention_streamk_min_tiles)


STREAMK_WAVES = 0         # 0 = automatic (8-wave workgroups where their plan applies, else 4); mavlm_set_attention_streamk_waves


def _streamk_plan_for(R: int, S: int, heads: int, waves: int):
    QB, G = 32 * waves, STREAMK_WGS * 4 // waves
    units = -(-R // QB) * heads
    if units <= G or -(-S // KV_TILE) < STREAMK_MIN_TILES:
        return 0, QB, 0, []
    rounds = -(-units // G)
    if units / (rounds * G) >= 0.95:
        return 0, QB, 0, []
    full, rem = divmod(units, G)
    base, levels = full * G, []
    for k in range(1, 5):
        if rem >= (G >> k):
            levels.append((k, base, G >> k))
            base += G >> k
            rem -= G >> k
    while rem > 0:
        n = min(rem, G >> 5)
        levels.append((5, base, n))
        base += n
        rem -= n
    return G, QB, full, levels


def streamk_plan(R: int, S: int, heads: int):
    """Mirrors attn3_plan (csrc/attention3.hip): (G, queries per unit, full, levels) with levels = [(k, first unit, units)]
    - after `full` whole units per workgroup the remaining units are cut into 2^k equal key ranges, level by level; G = 0:
    not used.  Units are 256-query blocks (8-wave workgroups, G = 256) where that plan applies, else 128 (G = 512)."""
    if STREAMK_WAVES != 4:
        p8 = _streamk_plan_for(R, S, heads, 8)
        if p8[0] or STREAMK_WAVES == 8:
            return p8
    return _streamk_plan_for(R, S, heads, 4)


STREAMK_AFFINE = True      # mirror of mavlm_set_attention_unit_order (1 = default: XCD-affine unit order)


def _xcd_first_unit(G: int, full: int, levels, x: int) -> int:
    W = G >> 3
    return x * full * W + sum(min(n, x * (W >> k)) for k, _, n in levels)


def streamk_unit_of_round(G: int, full: int, levels, si: int, v: int) -> int:
    """Mirrors attn3_unit_of_round (csrc/attention3.hip): the unit the v-th virtual workgroup runs in whole round si.  The
    units are dealt XCD-major: XCD x (virtual ids [x G/8, (x+1) G/8)) owns a contiguous range of the head-major unit order."""
    if not STREAMK_AFFINE:
        return si * G + v
    W = G >> 3
    x = v // W
    return _xcd_first_unit(G, full, levels, x) + si * W + (v - x * W)


def streamk_unit_of_level(G: int, full: int, levels, lv: int, ul: int) -> int:
    """Mirrors attn3_unit_of_level: the ul-th unit of level lv (the one cut into 2^k key ranges)."""
    if not STREAMK_AFFINE:
        return levels[lv][1] + ul
    W = G >> 3
    w = W >> levels[lv][0]
    x = ul // w
    u = _xcd_first_unit(G, full, levels, x) + full * W + (ul - x * w)
    for k, _, n in levels[:lv]:
        wj = W >> k
        u += min(max(n - x * wj, 0), wj)
    return u


def streamk_wgs(R: int, S: int, heads: int) -> int:
    return streamk_plan(R, S, heads)[0]


def streamk_split_tiles(R: int, S: int, heads: int):
    """(queries per unit, {(head, query block): [(tile_lo, tile_hi), ...]}) for the units the levelled stream-K schedule
    cuts: their keys are processed as 2^k ranges with independent online-softmax states and merged in key order like
    split-KV partials (empty ranges - fewer tiles than pieces - dropped).  Unit order: head-major, then query block."""
    G, QB, full, levels = streamk_plan(R, S, heads)
    out: Dict[Tuple[int, int], List[Tuple[int, int]]] = {}
    if not G:
        return QB, out
    nqb, nt = -(-R // QB), -(-S // KV_TILE)
    for lv, (k, base, n) in enumerate(levels):
        for ul in range(n):
            u = streamk_unit_of_level(G, full, levels, lv, ul)      # (XCD-affine unit order, round 4)
            rng = [((p * nt) >> k, ((p + 1) * nt) >> k) for p in range(1 << k)]
            out[(u // nqb, u % nqb)] = [(a, b) for a, b in rng if b > a]
    return QB, out


def frame_scores_fused(R: int, S: int, heads: int, patches: int, head_dim: int = 128) -> bool:
    """Mirrors mavlm_frame_scores_fused (csrc/mavlm_api.hip) for head_dim <= 128: does the fused step compute the frame
    scores inside the last formation layer's forward?  (Round 3: that launch runs the same schedule as the plain forward
    of the shape, so the answer no longer changes the rounding plan - kept for the tests of the C function.)"""
    return bool(FRAME_SCORES_FUSED and head_dim <= 128 and patches % 4 == 0 and patches >= KV_TILE and S > 0 and
                S % patches == 0 and S // patches <= 64 and heads * R * (64.0 + 31.0) * 8.0 < 2147483000.0 and
                split_plan(R, S, heads)[0] <= 1)          # (small grids keep their key splits)


def split_plan_wide(R: int, S: int, heads: int) -> Tuple[int, int]:
    """(number of key splits, tiles per split) of the wide-head kernels (32-key tiles, one 128-query workgroup per CU) - mirrors
    mavlm_attention_hd_splits (csrc/attention_hd.hip): small grids take the split count that minimises
    rounds(units x ns on 256 CUs) / ns + 2 % of a unit per split."""
    items = -(-R // 128) * heads
    nt = -(-S // 32)
    ns = 1
    if items < 200 and nt >= 32:
        cap = min(8, nt // 16)
        best = 5040
        for c in range(2, cap + 1):
            rounds = (items * c + 255) // 256
            score = rounds * 5040 // c + 100 * c
            if score < best:
                best, ns = score, c
    tps = -(-nt // ns)
    ns = -(-nt // tps)
    return ns, (tps if ns > 1 else 0)


WIDE_GROUPS = 2       # mirror of mavlm_set_attention_wide_groups: 2 (default) = 32-query waves at head_dim 448, 1 = 16-query


def streamk_plan_wide(R: int, S: int, heads: int):
    """Mirrors hd2_plan (csrc/attention_hd.hip): the levelled stream-K plan of the 32-query-wave kernel (head_dim 448) over
    128-query units, 32-key tiles and G = 256 workgroups; `heads` counts the heads of ALL videos of a row batch.  Returns
    (G, full, levels) with levels = [(k, first unit, units)]; G = 0: not used."""
    G = 256
    units = -(-R // 128) * heads
    if units <= G or -(-S // 32) < 2 * STREAMK_MIN_TILES:
        return 0, 0, []
    rounds = -(-units // G)
    if units / (rounds * G) >= 0.95:
        return 0, 0, []
    full, rem = divmod(units, G)
    base, levels = full * G, []
    for k in range(1, 5):
        if rem >= (G >> k):
            levels.append((k, base, G >> k))
            base += G >> k
            rem -= G >> k
    while rem > 0:
        n = min(rem, G >> 5)
        levels.append((5, base, n))
        base += n
        rem -= n
    return G, full, levels


def streamk_split_tiles_wide(R: int, S: int, heads: int):
    """{(global head, query block): [(tile_lo, tile_hi), ...]} for the 128-query units the wide-head stream-K plan cuts (32-key
    tiles; empty ranges dropped).  Unit order: head-major, then query block."""
    G, _, levels = streamk_plan_wide(R, S, heads)
    out: Dict[Tuple[int, int], List[Tuple[int, int]]] = {}
    if not G:
        return out
    nqb, nt = -(-R // 128), -(-S // 32)
    for k, base, n in levels:
        for ul in range(n):
            u = base + ul
            rng = [((p * nt) >> k, ((p + 1) * nt) >> k) for p in range(1 << k)]
            out[(u // nqb, u % nqb)] = [(a, b) for a, b in rng if b > a]
    return out


ROW_BLOCK_ELEMS = 1 << 28   # score elements per (head, query block) above which the queries are processed in blocks
LAZY_SCORE_ELEMS = 1 << 27  # ... and above which a block's scores are produced per 4096-key chunk (emulation modes)


# Heads and query blocks are independent: large problems spread them over a few threads (numpy releases the GIL inside
# its kernels).  Same operations on the same data in the same order per head / block - results do not depend on it.
FRAME_SCORES_FUSED = True        # mirror of mavlm_set_frame_score_mode (1 = default)
# Row batch (mavlm_config.batch, csrc/mavlm_api.hip): (b, B) = this computation is video b of B stepped together.  The
# arithmetic of a video does not change - only the attention SCHEDULE does: the stream-K plan is laid over B * heads
# "heads" (video-major), the small grids' split-KV form is not used.  (0, 1) = a single video.
ROW_BATCH = (0, 1)
PAR_THREADS = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
PAR_MIN_ELEMS = 1 << 22          # below this much work per call the threads cost more than they bring
_par_tls = threading.local()


def _par_map(fn, items):
    """[fn(x) for x in items], on a thread pool when there are several items and we are not already inside one"""
    if len(items) < 2 or PAR_THREADS < 2 or getattr(_par_tls, "busy", False):
        return [fn(x) for x in items]

    def guarded(x):
        _par_tls.busy = True
        try:
            return fn(x)
        finally:
            _par_tls.busy = False
    with ThreadPoolExecutor(max_workers=min(PAR_THREADS, len(items))) as ex:
        return list(ex.map(guarded, items))


def attention_heads(Q: np.ndarray, K: np.ndarray, V: np.ndarray, heads: int, mode: str = "fp32",
                    want_colsum: bool = False, want_probs: bool = False, kv_tile: Optional[int] = None,
                    wave_rows: Optional[int] = None, plain: bool = False):
    """Query-blocked driver of ``_attention_heads``: attention rows are independent (the wave-uniform rescale decision
    couples groups of ``wave_rows`` = 32 / 16 consecutive queries only, and the key-split plan depends on the TOTAL
    row count, which is passed down), so large problems (BASELINE configs[4]: 25 088 queries x 250 880 keys) run in
    blocks of 3136 queries instead of materialising a [R, Lk] score matrix per head; column sums add up over blocks."""
    R, Lk = Q.shape[0], K.shape[0]
    blk = 3136                                     # multiple of 32 and of 16
    if want_probs or R * Lk <= ROW_BLOCK_ELEMS or R <= blk:
        return _attention_heads(Q, K, V, heads, mode, want_colsum, want_probs, kv_tile, wave_rows, R, 0, plain)
    ctxs, lses, cs = [], [], None

    def one_block(r0):
        return _attention_heads(Q[r0:r0 + blk], K, V, heads, mode, want_colsum, False, kv_tile, wave_rows, R, r0, plain)
    for c_, l_, s_, _ in _par_map(one_block, list(range(0, R, blk))):      # (results in block order: sums stay deterministic)
        ctxs.append(c_)
        lses.append(l_)
        if want_colsum:
            cs = s_.astype(np.float64) if cs is None else cs + s_
    return (np.concatenate(ctxs, axis=0), np.concatenate(lses, axis=1), (cs.astype(F32) if want_colsum else None), None)


def _attention_heads(Q: np.ndarray, K: np.ndarray, V: np.ndarray, heads: int, mode: str = "fp32",
                     want_colsum: bool = False, want_probs: bool = False, kv_tile: Optional[int] = None,
                     wave_rows: Optional[int] = None, plan_rows: Optional[int] = None, row0: int = 0,
                     plain: bool = False):
    """softmax(Q K^T / sqrt(d)) V per head (MemoryController.py:51-54).  Returns
    (ctx [R,H*d] unrounded float32, lse2 [H,R] log2-domain log-sum-exp, colsum [H,Lk] | None, probs | None).

    fp32 mode: plain softmax.  Emulation modes follow the kernel's rounding points exactly: keys are consumed in
    tiles of ``kv_tile``; the probabilities of a tile are rounded to 16 bits *relative to the running reference
    value m at that tile* before the P.V product, while the row sum uses the unrounded values.  The reference m of a
    32-query wave moves to the running maximum only when some query of the wave saw its maximum grow by more than
    2^RESCALE_LOG2 (deferred rescale); earlier partial sums are then rescaled in float32."""
    r = rounder(mode)
    R, Lk = Q.shape[0], K.shape[0]
    d = Q.shape[1] // heads
    if kv_tile is None or wave_rows is None:
        kt_, wr_ = kernel_tiling(d)
        kv_tile = kv_tile or kt_
        wave_rows = wave_rows or wr_
    scale = F32(1.0 / math.sqrt(d))
    ctx = np.empty((R, heads * d), dtype=F32)
    lse2 = np.empty((heads, R), dtype=F32)
    colsum = np.zeros((heads, Lk), dtype=np.float64) if want_colsum else None
    probs = np.empty((heads, R, Lk), dtype=F32) if want_probs else None
    lazy = mode != "fp32" and not want_probs and R * Lk > LAZY_SCORE_ELEMS   # (eager is faster while [R, Lk] fits)

    def one_head(h):              # writes ctx[:, head], lse2[h], colsum[h], probs[h]: disjoint per head
        sl = slice(h * d, (h + 1) * d)
        if lazy:
            # large problems (emulation modes): scores are produced tile by tile instead of as one [R, Lk] matrix whose
            # 64-column slices would be strided by Lk * 4 bytes
            s = None
            Qh, Kh = np.ascontiguousarray(Q[:, sl]), np.ascontiguousarray(K[:, sl])
            chunk = {"k0": -1, "s": None}       # scores of the 4096-key chunk that holds the requested tile

            def tile_scores(k0, k1, Qh=Qh, Kh=Kh, chunk=chunk):
                c0 = (k0 // 4096) * 4096
                if chunk["k0"] != c0:
                    chunk["k0"], chunk["s"] = c0, _mm(Qh, Kh[c0:c0 + 4096].T) * scale
                if k1 - c0 > 4096:             # (a request that straddles chunks: the column-sum sweep is chunk-aligned,
                    return _mm(Qh, Kh[k0:k1].T) * scale     #  64-key tiles never straddle)
                return chunk["s"][:, k0 - c0:k1 - c0]
        else:
            s = _mm(Q[:, sl], K[:, sl].T) * scale                            # :51

            def tile_scores(k0, k1, s=s):
                return s[:, k0:k1]
        if mode == "fp32":
            m = s.max(axis=1, keepdims=True)
            p = np.exp(s - m, dtype=F32)
            l = p.sum(axis=1, keepdims=True, dtype=F32)
            acc = _mm(p, V[:, sl])
        else:
            def run(k_lo, k_hi, rows=None):
                """online softmax over the keys [k_lo, k_hi) for all queries, or for the index array `rows` (whole
                32-query waves, ascending)"""
                n = R if rows is None else len(rows)
                m = np.full((n, 1), -1e30, dtype=F32)
                l = np.zeros((n, 1), dtype=F32)
                acc = np.zeros((n, d), dtype=F32)
                for k0 in range(k_lo, k_hi, kv_tile):
                    st = tile_scores(k0, min(k0 + kv_tile, k_hi))
                    if rows is not None:
                        st = st[rows]
                    cand = np.maximum(m, st.max(axis=1, keepdims=True))
                    need = ((cand - m) * F32(1.4426950408889634) > F32(RESCALE_LOG2)).reshape(-1)
                    pad = (-n) % wave_rows
                    grp = np.concatenate([need, np.zeros(pad, bool)]).reshape(-1, wave_rows).any(axis=1)
                    move = np.repeat(grp, wave_rows)[:n].reshape(n, 1)
                    m_new = np.where(move, cand, m).astype(F32)
                    alpha = np.exp(m - m_new, dtype=F32)
                    pt = np.exp(st - m_new, dtype=F32)
                    l = l * alpha + pt.sum(axis=1, keepdims=True, dtype=F32)
                    acc = acc * alpha + _mm(r(pt), V[k0:min(k0 + kv_tile, k_hi), sl])   # P rounded as the MFMA operand
                    m = m_new
                return acc, m, l

            def merge(parts):
                """normalised fp32 partials + their log-sum-exp -> (acc, lse, 1): weights 2^(lse_s - lse), in order"""
                mx = np.maximum.reduce([p_[1] for p_ in parts])
                den = sum(np.exp(p_[1] - mx, dtype=F32) for p_ in parts)
                lse_t = (mx + np.log(den)).astype(F32)
                return sum(np.exp(p_[1] - lse_t, dtype=F32) * p_[0] for p_ in parts).astype(F32), lse_t

            sk_cuts = {}
            if plain:                         # the never-split grid (mavlm_attention; the fused step's last layer when
                ns, tps = 1, 0                # it carries the frame scores): one sweep over the keys for every row
            elif kv_tile == KV_TILE:          # attention3.hip (head_dim <= 128)
                vb, nb = ROW_BATCH
                ns, tps = split_plan(plan_rows or R, Lk, heads) if nb == 1 else (1, 0)
                sk_qb, sk_all = streamk_split_tiles(plan_rows or R, Lk, heads * nb)
                sk_cuts = {qb: a for (hh_, qb), a in sk_all.items() if hh_ == vb * heads + h}
                if streamk_wgs(plan_rows or R, Lk, heads * nb):
                    ns, tps = 1, 0
            elif kv_tile == 32:               # attention_hd.hip
                vb, nb = ROW_BATCH
                ns, tps = split_plan_wide(plan_rows or R, Lk, heads) if nb == 1 else (1, 0)
                if wave_rows == 32 and d == 448:      # attn_fwd_hd2_kernel: stream-K over the (video, head) pairs
                    sk_qb = 128
                    sk_all = streamk_split_tiles_wide(plan_rows or R, Lk, heads * nb)
                    sk_cuts = {qb: a for (hh_, qb), a in sk_all.items() if hh_ == vb * heads + h}
                    if streamk_plan_wide(plan_rows or R, Lk, heads * nb)[0]:
                        ns, tps = 1, 0
            else:
                ns, tps = 1, 0
            if sk_cuts:
                # levelled stream-K (attention3.hip, more units than workgroup slots, long key sequences): a cut unit is
                # computed as 2^k key ranges with fresh softmax states, merged in key order; all other rows see their keys
                # in one sweep
                cut = np.zeros(R, dtype=np.int64)              # 0 = not cut, else 1 + index into `plans`
                plans = []
                for qb, rng in sk_cuts.items():
                    lo_, hi_ = qb * sk_qb - row0, (qb + 1) * sk_qb - row0
                    if hi_ <= 0 or lo_ >= R:
                        continue
                    if rng not in plans:
                        plans.append(rng)
                    cut[max(lo_, 0):min(hi_, R)] = 1 + plans.index(rng)
                acc = np.empty((R, d), dtype=F32)
                m = np.empty((R, 1), dtype=F32)
                l = np.ones((R, 1), dtype=F32)
                for pi in np.unique(cut):
                    rows = np.nonzero(cut == pi)[0]
                    if pi == 0:
                        ac_, m_, l_ = run(0, Lk, rows)
                        acc[rows], m[rows] = (ac_ / l_).astype(F32), (m_ + np.log(l_)).astype(F32)
                    else:
                        parts = []
                        for t_lo, t_hi in plans[pi - 1]:
                            ac_, m_, l_ = run(t_lo * kv_tile, min(t_hi * kv_tile, Lk), rows)
                            parts.append(((ac_ / l_).astype(F32), (m_ + np.log(l_)).astype(F32)))
                        acc[rows], m[rows] = merge(parts)
            elif ns == 1:
                acc, m, l = run(0, Lk)
            else:
                # split-KV (attention3.hip, small grids): each split yields a normalised fp32 partial + its
                # log-sum-exp; the merge weights them by 2^(lse_s - lse)
                parts = []
                for sp in range(ns):
                    a_, m_, l_ = run(sp * tps * kv_tile, min((sp + 1) * tps * kv_tile, Lk))
                    parts.append(((a_ / l_).astype(F32), (m_ + np.log(l_)).astype(F32)))
                acc, m = merge(parts)
                l = np.ones((R, 1), dtype=F32)
        ctx[:, sl] = acc / l                                                 # :53
        lse = m + np.log(l)
        lse2[h] = (lse / F32(math.log(2.0))).reshape(-1)
        if lazy and want_colsum:
            for k0 in range(0, Lk, 4096):
                k1 = min(k0 + 4096, Lk)
                colsum[h, k0:k1] = np.exp(tile_scores(k0, k1) - lse, dtype=F32).sum(axis=0, dtype=np.float64)
        elif want_colsum or want_probs:
            pn = np.exp(s - lse, dtype=F32)                                  # :52 normalised probabilities
            if want_colsum:
                colsum[h] = pn.sum(axis=0, dtype=np.float64)
            if want_probs:
                probs[h] = pn
    if heads > 1 and R * Lk >= PAR_MIN_ELEMS:
        _par_map(one_head, list(range(heads)))
    else:
        for h in range(heads):
            one_head(h)
    return ctx, lse2, (colsum.astype(F32) if want_colsum else None), probs


def mha(Xq: np.ndarray, Xkv: np.ndarray, w: Dict[str, np.ndarray], prefix: str, cfg: PathConfig,
        mode: str = "fp32", want_colsum: bool = False, want_probs: bool = False,
        kv_cached: Optional[Tuple[np.ndarray, np.ndarray]] = None):
    """``Attention.forward`` (MemoryController.py:47-57): q/k/v Linear, H heads, softmax(qk^T/sqrt d),
    P V, merge heads, Residual = LN(dense(ctx) + Xq).  Returns (out[R,D], colsum[Lk] | None, probs | None).

    Emulation mode rounds: Q,K,V (stored), P (MFMA operand, see attention_heads), ctx (stored), LN output
    (stored).  dense+bias+residual stays float32 into the LN.
    """
    r = rounder(mode)
    Q = r(linear(Xq, w[f"{prefix}.q_proj.weight"], w[f"{prefix}.q_proj.bias"]))
    if kv_cached is None:
        K = r(linear(Xkv, w[f"{prefix}.k_proj.weight"], w[f"{prefix}.k_proj.bias"]))
        V = r(linear(Xkv, w[f"{prefix}.v_proj.weight"], w[f"{prefix}.v_proj.bias"]))
    else:
        K, V = kv_cached
    # (the fused step computes the frame scores inside this attention when it can, on the schedule every attention of the
    # shape runs: asking for the scores does not change the context)
    ctx, _, colsum_h, probs = attention_heads(Q, K, V, cfg.heads, mode, want_colsum, want_probs)
    colsum = colsum_h.astype(np.float64).sum(axis=0).astype(F32) if want_colsum else None   # :135 sum over heads
    ctx = r(ctx)
    pre = linear(ctx, w[f"{prefix}.residual.dense.weight"], w[f"{prefix}.residual.dense.bias"]) + Xq
    out = r(layernorm(pre, w[f"{prefix}.residual.layernorm.weight"],
                      w[f"{prefix}.residual.layernorm.bias"], cfg.eps))     # :26-29,55
    return out, colsum, probs


def project_kv(X: np.ndarray, w: Dict[str, np.ndarray], prefix: str, mode: str):
    r = rounder(mode)
    return (r(linear(X, w[f"{prefix}.k_proj.weight"], w[f"{prefix}.k_proj.bias"])),
            r(linear(X, w[f"{prefix}.v_proj.weight"], w[f"{prefix}.v_proj.bias"])))


def transformer_layer(m: np.ndarray, seg: np.ndarray, w: Dict[str, np.ndarray], prefix: str,
                      cfg: PathConfig, mode: str, want_colsum: bool):
    """``TransformerLayer.forward`` (MemoryController.py:69-72): cross-attn -> Linear(D,4D)+ReLU ->
    Residual(4D->D)."""
    r = rounder(mode)
    a, colsum, _ = mha(m, seg, w, f"{prefix}.memory_segment_fusion_attention", cfg, mode, want_colsum)
    h = r(np.maximum(linear(a, w[f"{prefix}.mlp.0.weight"], w[f"{prefix}.mlp.0.bias"]), 0))
    pre = linear(h, w[f"{prefix}.residual.dense.weight"], w[f"{prefix}.residual.dense.bias"]) + a
    out = r(layernorm(pre, w[f"{prefix}.residual.layernorm.weight"],
                      w[f"{prefix}.residual.layernorm.bias"], cfg.eps))
    return out, colsum


# --------------------------------------------------------------------------- #
# a8-a10  recurrent memory transformer
# --------------------------------------------------------------------------- #
@dataclass
class RecurrentMemory:
    """``TransformerProjector`` (MemoryController.py:74-158) restated.  ``memory_cache`` and
    ``frame_attn_scores`` mirror the reference attributes (the latter is never cleared by the
    reference either, MemoryController.py:157)."""
    cfg: PathConfig
    w: Dict[str, np.ndarray]
    mode: str = "fp32"
    prefix: str = "recurrent_memory_transformer"
    frame_scores: bool = True
    memory_cache: List[np.ndarray] = field(default_factory=list)
    frame_attn_scores: List[np.ndarray] = field(default_factory=list)
    _kv_cache: List[Tuple[np.ndarray, np.ndarray]] = field(default_factory=list)

    def reset(self):
        """llava_arch.py:532 - the caller clears memory_cache per video."""
        self.memory_cache = []
        self._kv_cache = []

    def initial(self) -> np.ndarray:
        r = rounder(self.mode)
        cfg = self.cfg
        m = r(self.w[f"{self.prefix}.initial_memory"] + self.w[f"{self.prefix}.memory_pos_embed"])  # :123-124
        return m.reshape(cfg.mem_rows, cfg.hidden)

    def evolve(self, last: np.ndarray) -> np.ndarray:
        """``_update_memory_tokens_with_cache`` (:89-115): q = last memory, kv = cat(cache) (which
        includes ``last``).  K/V of each cached memory are projected once and kept (row-independent,
        exact).  The per-chunk statistics at :99-109 are dead code and are not restated."""
        cfg = self.cfg
        pfx = f"{self.prefix}.memory_update_attention"
        while len(self._kv_cache) < len(self.memory_cache):
            mem = self.memory_cache[len(self._kv_cache)].reshape(cfg.mem_rows, cfg.hidden)
            self._kv_cache.append(project_kv(mem, self.w, pfx, self.mode))
        K = np.concatenate([k for k, _ in self._kv_cache], axis=0)
        V = np.concatenate([v for _, v in self._kv_cache], axis=0)
        out, _, _ = mha(last, None, self.w, pfx, cfg, self.mode, kv_cached=(K, V))
        return out

    def step(self, seg: np.ndarray):
        """``TransformerProjector.forward`` (:118-158).  seg: [F,P,D] (already PE-added)."""
        cfg = self.cfg
        F_, P, D = seg.shape
        assert P == cfg.patches and D == cfg.hidden
        r = rounder(self.mode)
        if self.memory_cache:
            m = self.evolve(self.memory_cache[-1].reshape(cfg.mem_rows, D))        # :125-127
        else:
            m = self.initial()
        x = r(seg.reshape(F_ * P, D))
        colsum = None
        for l in range(cfg.depth):                                                   # :132-133
            last = l == cfg.depth - 1
            m, cs = transformer_layer(m, x, self.w, f"{self.prefix}.layers.{l}", cfg, self.mode,
                                      want_colsum=(last and self.frame_scores))
            if last:
                colsum = cs
        if colsum is not None:
            scores = colsum.reshape(F_, P).mean(axis=1).astype(F32)                 # :135-139
            self.frame_attn_scores.append(scores)                                    # :156-157
        self.memory_cache.append(m.reshape(cfg.mem_tokens, P, D))                    # :152
        if len(self.memory_cache) > cfg.cache_cap:                                   # :153-154
            drop = len(self.memory_cache) - cfg.cache_cap
            self.memory_cache = self.memory_cache[drop:]
            self._kv_cache = self._kv_cache[drop:]
        return self.memory_cache, self.frame_attn_scores


# --------------------------------------------------------------------------- #
# a11-a13  fuser MLP, token-type add, concat
# --------------------------------------------------------------------------- #
def fuser_mlp(x: np.ndarray, w: Dict[str, np.ndarray], mode: str = "fp32",
              add: Optional[np.ndarray] = None) -> np.ndarray:
    """``memory_fuser`` = Linear(D,4D) -> GELU(erf) -> Linear(4D,D) (llava_arch.py:132-136,546),
    no residual, no norm; ``add`` = token_type_embedding[0] fused into the second epilogue (:548-553)."""
    r = rounder(mode)
    shp = x.shape
    x2 = r(x.reshape(-1, shp[-1]))
    u = r(gelu_erf(linear(x2, w["memory_fuser.0.weight"], w["memory_fuser.0.bias"])))
    y = linear(u, w["memory_fuser.2.weight"], w["memory_fuser.2.bias"])
    if add is not None:
        y = y + add.astype(F32)
    return r(y).reshape(shp)


def video_tokens(x: np.ndarray, frame_idx: np.ndarray, cfg: PathConfig, w: Dict[str, np.ndarray],
                 embed_tokens: np.ndarray, mode: str = "fp32", frame_scores: bool = True,
                 return_parts: bool = False):
    """The per-video memory driver, llava_arch.py:502-557 + 613-629 + 705-731:
    PE add -> fine frames -> chunk loop -> fuser -> type add -> [mem_prompt ; memory ; newline ;
    frame_prompt ; fine ; newline].  ``x``: pooled frame tokens [T,P,D]; ``embed_tokens``: [vocab,D]
    table of the host LLM (only the 19 prompt ids are read)."""
    r = rounder(mode)
    T = x.shape[0]
    xp = pe_add(x, frame_idx, w["positional_encoding.frame_embed"], mode)          # :510-511
    fine = xp[fine_frame_indices(T, cfg.fine_frames)]                               # :513-524
    rm = RecurrentMemory(cfg, w, mode, frame_scores=frame_scores)
    rm.reset()                                                                      # :532
    b = uniform_segment_variant(T, cfg.chunk)                                       # :528
    for i in range(len(b) - 1):                                                     # :534-537
        rm.step(xp[b[i]:b[i + 1]])
    mem = np.concatenate(rm.memory_cache, axis=0)                                   # :545
    E = r(w["token_type_embedding.weight"])
    mem = fuser_mlp(mem, w, mode, add=E[0])                                         # :546-553
    fine = r(fine + E[1])                                                           # :554
    nl = r(w["image_newline"])[None]
    et = r(embed_tokens)
    toks = np.concatenate([et[MEM_PROMPT_IDS], mem.reshape(-1, cfg.hidden), nl,
                           et[FRAME_PROMPT_IDS], fine.reshape(-1, cfg.hidden), nl], axis=0)  # :620-629,729-731
    if return_parts:
        return toks, dict(pe=xp, memory=rm.memory_cache, fused=mem, fine=fine,
                          frame_scores=rm.frame_attn_scores)
    return toks


# --------------------------------------------------------------------------- #
# a14  splice into the text sequence
# --------------------------------------------------------------------------- #
IGNORE_INDEX = -100          # llava/constants.py:7
IMAGE_TOKEN_INDEX = -200     # llava/constants.py:8


def splice(input_ids: np.ndarray, labels: Optional[np.ndarray], attention_mask: Optional[np.ndarray],
           image_tokens: np.ndarray, embed_tokens: np.ndarray, max_len: Optional[int] = None,
           padding_side: str = "right"):
    """Batch-1..B splice of the per-video token block at IMAGE_TOKEN_INDEX, truncate, pad
    (llava_arch.py:745-866).  Every sample uses ``image_tokens`` in order of appearance; the
    reference supports batch size 1 for the memory path (:436).  Returns
    (embeds[B,L,D], labels[B,L], mask[B,L] bool, position_ids[B,L])."""
    B = input_ids.shape[0]
    if attention_mask is None:
        attention_mask = np.ones_like(input_ids, dtype=bool)
    if labels is None:
        labels = np.full_like(input_ids, IGNORE_INDEX)
    outs, labs = [], []
    for b in range(B):
        ids = input_ids[b][attention_mask[b].astype(bool)]
        lab = labels[b][attention_mask[b].astype(bool)]
        pieces, lpieces = [], []
        start = 0
        pos = list(np.where(ids == IMAGE_TOKEN_INDEX)[0]) + [len(ids)]
        for j, p in enumerate(pos):
            pieces.append(embed_tokens[ids[start:p]])
            lpieces.append(lab[start:p])
            if j < len(pos) - 1:
                pieces.append(image_tokens)
                lpieces.append(np.full(image_tokens.shape[0], IGNORE_INDEX, dtype=lab.dtype))
            start = p + 1
        e = np.concatenate(pieces, axis=0)[:max_len]
        l = np.concatenate(lpieces, axis=0)[:max_len]
        outs.append(e)
        labs.append(l)
    L = max(e.shape[0] for e in outs)
    D = image_tokens.shape[1]
    emb = np.zeros((B, L, D), dtype=F32)
    lab = np.full((B, L), IGNORE_INDEX, dtype=np.int64)
    msk = np.zeros((B, L), dtype=bool)
    pid = np.zeros((B, L), dtype=np.int64)
    for b, (e, l) in enumerate(zip(outs, labs)):
        n = e.shape[0]
        sl = slice(L - n, L) if padding_side == "left" else slice(0, n)
        if n:
            emb[b, sl] = e
            lab[b, sl] = l
            msk[b, sl] = True
            pid[b, sl] = np.arange(n)
    return emb, lab, msk, pid


# --------------------------------------------------------------------------- #
# a16  step before the path: bilinear 2x2 pool (next-row)
# --------------------------------------------------------------------------- #
def bilinear_pool(x: np.ndarray, side: int = 27, stride: int = 2) -> np.ndarray:
    """``get_2dPool`` bilinear branch (llava_arch.py:277-297): [F,side*side,D] ->
    F.interpolate(size=ceil(side/stride), mode='bilinear', align_corners=False) -> [F,out*out,D]."""
    Fn, N, D = x.shape
    assert N == side * side
    out = math.ceil(side / stride)
    img = x.reshape(Fn, side, side, D).astype(F32)
    scale = F32(side) / F32(out)

    def src(n_out):
        c = (np.arange(n_out, dtype=F32) + F32(0.5)) * scale - F32(0.5)
        c = np.maximum(c, F32(0.0))
        i0 = np.floor(c).astype(np.int64)
        i1 = np.minimum(i0 + 1, side - 1)
        lam = (c - i0.astype(F32)).astype(F32)
        return i0, i1, lam

    y0, y1, ly = src(out)
    x0, x1, lx = src(out)
    top = img[:, y0][:, :, x0] * (1 - lx)[None, None, :, None] + img[:, y0][:, :, x1] * lx[None, None, :, None]
    bot = img[:, y1][:, :, x0] * (1 - lx)[None, None, :, None] + img[:, y1][:, :, x1] * lx[None, None, :, None]
    res = top * (1 - ly)[None, :, None, None] + bot * ly[None, :, None, None]
    return res.reshape(Fn, out * out, D).astype(F32)


# --------------------------------------------------------------------------- #
# algorithmic work - used by bench.py for the roofline line
# --------------------------------------------------------------------------- #
def flops_formation(R: int, S: int, D: int, L: int) -> float:
    return L * (20.0 * R * D * D + 4.0 * S * D * D + 4.0 * R * S * D)


def flops_evolution(R: int, C: int, D: int) -> float:
    """K/V of cached memories projected once per memory: 4*R*D^2 (q,out) + 4*R*D^2 (new K/V) + attn."""
    return 4.0 * R * D * D + 4.0 * R * D * D + 4.0 * R * C * D


def flops_fuser(N: int, D: int) -> float:
    return 16.0 * N * D * D


def rel_l2(a: np.ndarray, b: np.ndarray) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
