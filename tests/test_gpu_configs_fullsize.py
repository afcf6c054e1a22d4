"""-m gpu: BASELINE.json configs[3] and configs[4] at their STATED sizes on one MI355X.

configs[4]: 512-frame videos, 128 memory tokens, fp16 MFMA path - 16 chunks of 32 frames, the FIFO (cap 10) saturates
            and wraps; the evolution attention runs 25 088 queries against 250 880 keys (SURVEY.md §7 "hard parts").
configs[3]: a 1024-frame video at the LLaVA-OneVision-7B width (D = 3584, 8 heads of 448): 32 chunks, three FIFO wraps.
            (Its 8-GPU row-sharded form is covered at this width in test_gpu_distributed.py.)

Both are far too large for the CPU oracle at full width, so each config has two tests:
  * full size: size-independent properties (finite, sum_f score_f = H*R/P on every chunk, LayerNorm moments of every
    cached memory, determinism, eager launches == hipGraph replay bit for bit);
  * the same ROW COUNT / FIFO behaviour against the oracle at a width the host can afford (the kernels' grids, key
    counts per step and ring arithmetic are those of the full configuration; only the hidden width shrinks).
"""
import numpy as np
import pytest
import torch

import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd.model import llava_arch as arch
from oracle import memory_path as O
from gpu_util import to_dev, to_np
from test_gpu_path import TOL, _tiny_host, chain_tol, make_projector, run_oracle_steps

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_path():
    with torch.no_grad():
        yield


def _randn16(shape, seed, dtype):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda", dtype=torch.float32).to(dtype)


def _layernorm_moments(mem, ln):
    z = (mem.float() - ln.bias.float()) / ln.weight.float()
    return float(z.mean(-1).abs().max()), float((z.var(-1, unbiased=False) - 1).abs().max())


def test_config4_full_size_512_frames_128_memory_tokens_fp16():
    """configs[4] on one GPU (each of the 8 GPUs runs exactly this): T = 512, M = 128, D = 1024, fp16."""
    T, M, D, H = 512, 128, 1024, 8
    cfg = O.PathConfig(hidden=D, heads=H, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=41, grid="fp16")
    model, _ = _tiny_host(cfg, w, "fp16")
    rm = model.recurrent_memory_transformer
    x = _randn16((T, 196, D), 4100, torch.float16)
    idx = torch.arange(T)
    mp = _randn16((10, D), 4101, torch.float16) * 0.02
    fp = _randn16((9, D), 4102, torch.float16) * 0.02
    n0 = len(rm.frame_attn_scores)
    eager, info = arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    assert info["num_memories"] == 10 and len(rm.memory_cache) == 10            # FIFO saturated (16 chunks, cap 10)
    assert eager.shape == (arch.video_token_rows(T, M), D) and eager.dtype == torch.float16
    assert torch.isfinite(eager.float()).all()
    scores = rm.frame_attn_scores[n0:]
    assert len(scores) == 16
    for s_ in scores:                                                             # MemoryController.py:135-139
        assert s_.shape == (32,) and abs(float(s_.float().sum()) - H * M) < 0.02 * H * M
    ln = rm.layers[1].residual.layernorm
    for mem in rm.memory_cache:                                                   # every cached memory is a LayerNorm output
        mean_err, var_err = _layernorm_moments(mem, ln)
        assert mean_err < 2e-2 and var_err < 5e-2, (mean_err, var_err)
    # the ring wrapped: the cache is oldest-first, i.e. memories of chunks 6..15 sit in slots 6,7,8,9,0,...,5
    eng = rm.engine(x.device, x.dtype)
    for i, mem in enumerate(rm.memory_cache):
        assert mem.data_ptr() == eng.mem_ring[(6 + i) % 10].data_ptr()
    again, _ = arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
    assert torch.equal(again, eager)                                              # deterministic, state fully reset
    g = arch.GraphedVideoMemory(model, T, idx)
    out = g(x, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    assert torch.equal(out, eager)                                                # hipGraph replay == eager launches


def test_config4_row_count_vs_oracle_fp16_ring_wraps():
    """The configs[4] memory (M = 128: 25 088 rows, 196 query blocks) against the oracle at H = 1, D = 128, fp16, FIFO
    cap 2 over three one-frame chunks: the ring wraps (chunk 2 overwrites slot 0) and the evolution attends over 50 176
    keys.  fp16 chains stay under the flat 1e-3 gate (no noise-floor allowance).  (~1 minute of host time: the oracle
    walks 25 088 x 50 176 scores in 64-key tiles.)"""
    cfg = O.PathConfig(hidden=128, heads=1, mem_tokens=128, depth=2, cache_cap=2)
    w = O.make_weights(cfg, seed=42, grid="fp16")
    proj = make_projector(cfg, w, "fp16", cache_cap=2)
    segs = [O.fp16_round(O.hash_normal_like((1, 196, 128), 4200 + t)) for t in range(3)]
    ref = run_oracle_steps(cfg, w, "fp16", segs, np.float32)
    proj.memory_cache = []
    for t, seg in enumerate(segs):
        cache, scores = proj(to_dev(seg, "fp16"))
        assert len(cache) == len(ref[t][0]) == min(t + 1, 2)
        for i in range(len(cache)):
            err = O.rel_l2(to_np(cache[i]), ref[t][0][i])
            assert err < TOL, (t, i, err)
        assert O.rel_l2(to_np(scores[-1]), ref[t][1]) < 5e-3
        assert abs(float(scores[-1].float().sum()) - 128) < 0.02 * 128


def test_config3_full_size_1024_frames_ov7b_width():
    """configs[3] frame count AND width on one GPU: T = 1024 (32 chunks, three FIFO wraps), D = 3584, 8 heads of 448,
    the reference's 8 memory tokens, bf16; needs `memory_max_frames >= 1024` (the reference's table stops at 600)."""
    T, M, D, H = 1024, 8, 3584, 8
    cfg = O.PathConfig(hidden=D, heads=H, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=43)
    model, hf = _tiny_host(cfg, w)
    rm = model.recurrent_memory_transformer
    from memory_augmented_vlm_amd.model.memory_module.position_encoding import TemporalPositionalEncoding
    with pytest.raises(ValueError):                                               # position_encoding.py:73-74
        model.positional_encoding.check_indices(torch.arange(T))
    model.positional_encoding = TemporalPositionalEncoding(max_frames=T, embed_dim=D, learnable=False).to("cuda").to(torch.bfloat16)
    x = _randn16((T, 196, D), 4300, torch.bfloat16)
    idx = torch.arange(T)
    mp = _randn16((10, D), 4301, torch.bfloat16) * 0.02
    fp = _randn16((9, D), 4302, torch.bfloat16) * 0.02
    n0 = len(rm.frame_attn_scores)
    eager, info = arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    assert info["num_memories"] == 10 and eager.shape == (arch.video_token_rows(T, M), D)
    assert torch.isfinite(eager.float()).all()
    scores = rm.frame_attn_scores[n0:]
    assert len(scores) == 32
    for s_ in scores:
        assert abs(float(s_.float().sum()) - H * M) < 0.02 * H * M
    ln = rm.layers[1].residual.layernorm
    for mem in rm.memory_cache:
        mean_err, var_err = _layernorm_moments(mem, ln)
        assert mean_err < 2e-2 and var_err < 5e-2, (mean_err, var_err)
    g = arch.GraphedVideoMemory(model, T, idx)
    out = g(x, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_config3_width_chain_vs_oracle_ring_wraps():
    """The OV-7B width (D = 3584, wide-head kernels) against the oracle along a chain long enough for the FIFO
    (cap 2) to wrap: 4 one-frame chunks, 8 memory tokens.  Gate = the calibrated chain tolerance."""
    cfg = O.PathConfig(hidden=3584, heads=8, mem_tokens=8, depth=2, cache_cap=2)
    w = O.make_weights(cfg, seed=44)
    proj = make_projector(cfg, w, "bf16", cache_cap=2)
    segs = [O.bf16_round(O.hash_normal_like((f, 196, 3584), 4400 + t)) for t, f in enumerate((1, 1, 1, 1))]
    ref = run_oracle_steps(cfg, w, "bf16", segs, np.float32)
    alt = run_oracle_steps(cfg, w, "bf16", segs, np.float64)
    proj.memory_cache = []
    for t, seg in enumerate(segs):
        cache, scores = proj(to_dev(seg))
        assert len(cache) == min(t + 1, 2)
        floor = O.rel_l2(alt[t][0][-1], ref[t][0][-1])
        err = O.rel_l2(to_np(cache[-1]), ref[t][0][-1])
        print(f"D=3584 step {t}: HIP vs oracle {err:.2e} (floor {floor:.2e})")
        assert err < chain_tol(floor), (t, err, floor)
        assert O.rel_l2(to_np(scores[-1]), ref[t][1]) < 5e-3
