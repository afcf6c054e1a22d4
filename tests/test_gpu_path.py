"""-m gpu: parity of the fused recurrent-memory path (mavlm_step / mavlm_fuse_emit through the reference-shaped
modules) against the CPU oracle in emulation mode, against golden vectors of the reference itself (G7), and
size-independent properties at the BASELINE sizes."""
import math
import types

import numpy as np
import pytest
import torch

import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi
from memory_augmented_vlm_amd.model.memory_module.MemoryController import Config, TransformerProjector
from memory_augmented_vlm_amd.model.memory_module.position_encoding import TemporalPositionalEncoding
from memory_augmented_vlm_amd.model import llava_arch as arch
from oracle import memory_path as O
from conftest import load_golden
from gpu_util import to_dev, to_np, load_oracle_weights, DT

pytestmark = pytest.mark.gpu
TOL = 1e-3            # GPU vs oracle in operand-rounding-emulation mode (primary gate)
TOL_REF_FP32 = 1.2e-2  # GPU bf16 vs the reference's own fp32 run; the reference's bf16 run sits at 4.8e-3..8.4e-3


@pytest.fixture(autouse=True)
def _inference_path():
    """This file tests the inference path; with grad enabled the modules take the autograd (training) path, which
    tests/test_gpu_backward_path.py covers."""
    with torch.no_grad():
        yield


def chain_tol(floor):
    """Tolerance for a CHAIN of 16-bit-rounded stages.  Single stages match the oracle to ~1e-4 (see
    test_gpu_ops.py and test_step_stagewise_teacher_forced); along a chain two correct implementations
    decorrelate at every storage rounding (a perturbation d << ulp becomes ~sqrt(d*ulp)), so the gate is the larger
    of 1e-3 and twice the measured noise floor = the oracle against itself with float64 accumulation."""
    return max(TOL, 2.0 * floor)


def run_oracle_steps(cfg, w, mode, segs, acc, cap=None):
    with O.accumulate_in(acc):
        rm = O.RecurrentMemory(cfg, w, mode)
        rm.reset()
        outs = []
        for seg in segs:
            cache, scores = rm.step(seg)
            outs.append(([c.copy() for c in cache], scores[-1].copy()))
    return outs


def make_projector(cfg: O.PathConfig, w, mode="bf16", cache_cap=10):
    c = Config()
    c.mm_hidden_size = cfg.hidden
    c.mm_intermediate_size = cfg.inter
    c.mm_num_attention_heads = cfg.heads
    c.num_memory_tokens = cfg.mem_tokens
    c.patch_size = cfg.patches
    c.depth = cfg.depth
    c.mm_layer_norm_eps = cfg.eps
    c.mm_dtype = torch.float32
    c.cache_cap = cache_cap
    m = TransformerProjector(c).eval()
    pfx = "recurrent_memory_transformer."
    n = load_oracle_weights(m, {k[len(pfx):]: v for k, v in w.items() if k.startswith(pfx)})
    assert n == len(m.state_dict())
    return m.to("cuda").to(DT[mode])


@pytest.mark.parametrize("mode,H,M,frames,hd", [("bf16", 8, 8, [3, 2, 3, 1], 128), ("fp16", 2, 4, [2, 1, 2], 128),
                                                 ("bf16", 1, 3, [1, 1], 128), ("bf16", 8, 8, [2, 3, 1], 112),
                                                 ("fp16", 4, 2, [1, 2], 64), ("bf16", 2, 4, [2, 1, 2], 448),
                                                 ("bf16", 8, 8, [2, 1], 448)])
def test_recurrent_steps_vs_oracle(mode, H, M, frames, hd):
    """hd = 112 is the Qwen2-0.5B shape (D = 896, the reference Config default): heads run zero-padded to 128.
    hd = 448 is the LLaVA-OneVision-7B shape (D = 3584 with 8 heads): wide-head kernels, attention_hd.hip."""
    cfg = O.PathConfig(hidden=hd * H, heads=H, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=5, grid=mode)
    proj = make_projector(cfg, w, mode)
    r = O.rounder(mode)
    segs = [r(O.hash_normal_like((f, 196, cfg.hidden), 900 + t)) for t, f in enumerate(frames)]
    ref = run_oracle_steps(cfg, w, mode, segs, np.float32)
    alt = run_oracle_steps(cfg, w, mode, segs, np.float64)
    proj.memory_cache = []
    with torch.no_grad():
        for t, seg in enumerate(segs):
            cache, scores = proj(to_dev(seg, mode))
            ocache, oscores = ref[t]
            assert len(cache) == len(ocache) == t + 1
            floor = O.rel_l2(alt[t][0][-1], ocache[-1])
            err = O.rel_l2(to_np(cache[-1]), ocache[-1])
            print(f"{mode} H={H} step {t}: HIP vs oracle {err:.2e}, oracle f64-acc vs f32-acc (floor) {floor:.2e}")
            assert err < chain_tol(floor), (t, err, floor)
            if mode == "fp16":
                assert err < TOL        # the finer grid keeps the whole chain under the flat 1e-3 gate
            for i in range(len(cache) - 1):       # older entries are untouched views of the ring
                assert O.rel_l2(to_np(cache[i]), ref[i][0][-1]) < chain_tol(O.rel_l2(alt[i][0][-1], ref[i][0][-1]))
            assert tuple(cache[-1].shape) == (M, 196, cfg.hidden)
            assert O.rel_l2(to_np(scores[-1]), oscores) < 5e-3   # scores are stored in the 16-bit model dtype
            # property (MemoryController.py:135-139): sum_f score_f = H * R / P
            assert abs(float(scores[-1].float().sum()) - H * M) < 2e-2 * H * M


@pytest.mark.parametrize("score_mode", [1, 0])
def test_step_stagewise_teacher_forced(score_mode, request):
    """Every stage INSIDE the fused mavlm_step against the oracle fed with the GPU's own inputs to that stage
    (read back from the workspace after the step).  Tight gate: <= 1e-3 per stage, expected ~1e-4.  Both ways of getting
    the frame scores: 1 = fused into the last layer's forward (default; the per-key column sums are then never formed), 0 =
    the column-sum pass (its [H, S] result is checked as a stage of its own)."""
    capi.check(capi.lib().mavlm_set_frame_score_mode(score_mode), "frame score mode")
    request.addfinalizer(lambda: capi.lib().mavlm_set_frame_score_mode(1))
    # (the fp32 dense output `pre` is a stage of its own here: keep the split-K reduction and the LayerNorm two kernels - the default
    #  one-kernel form never writes it; test_splitk_reduction_inside_the_layernorm_kernel_same_bits compares the two forms)
    capi.check(capi.lib().mavlm_set_splitk_layernorm(0), "splitk layernorm")
    request.addfinalizer(lambda: capi.lib().mavlm_set_splitk_layernorm(1))
    H, M, mode = 8, 8, "bf16"
    cfg = O.PathConfig(hidden=1024, heads=H, mem_tokens=M, depth=2)
    D, R = cfg.hidden, cfg.mem_rows
    w = O.make_weights(cfg, seed=12)
    proj = make_projector(cfg, w, mode)
    r = O.bf16_round
    segs = [r(O.hash_normal_like((f, 196, D), 1200 + t)) for t, f in enumerate((3, 2))]
    proj.memory_cache = []
    with torch.no_grad():
        for seg in segs:
            cache, scores = proj(to_dev(seg))
    torch.cuda.synchronize()
    eng = proj.engine(torch.device("cuda", 0), torch.bfloat16)
    ws = {k: to_np(v) if v.dtype != torch.float32 else v.cpu().numpy() for k, v in eng.workspace_views().items()}
    S = 2 * 196
    x = segs[1].reshape(S, D)
    T = "recurrent_memory_transformer"
    errs = {}

    def lin(xin, name):
        return O.linear(xin, w[f"{name}.weight"], w[f"{name}.bias"])

    # chunk K/V for both layers (one GEMM on the GPU)
    kv = ws["kv_seg"][:S]
    for l in range(2):
        a_ = f"{T}.layers.{l}.memory_segment_fusion_attention"
        errs[f"K{l}"] = O.rel_l2(kv[:, (2 * l) * D:(2 * l + 1) * D], r(lin(x, a_ + ".k_proj")))
        errs[f"V{l}"] = O.rel_l2(kv[:, (2 * l + 1) * D:(2 * l + 2) * D], r(lin(x, a_ + ".v_proj")))
    # evolution (t=1): newest = cache[0]; its K|V projection sits in ring slot 0; output = mA
    newest = to_np(cache[0]).reshape(R, D)
    evo = f"{T}.memory_update_attention"
    ekv = to_np(eng.evo_kv[0])
    errs["evoK"] = O.rel_l2(ekv[:, :D], r(lin(newest, evo + ".k_proj")))
    errs["evoV"] = O.rel_l2(ekv[:, D:], r(lin(newest, evo + ".v_proj")))
    mA, _, _ = O.mha(newest, None, w, evo, cfg, mode, kv_cached=(ekv[:, :D], ekv[:, D:]))
    errs["evolved(q,attn,dense,LN)"] = O.rel_l2(ws["mA"], mA)
    # last formation layer (l=1): its input is mB (layer-0 output)
    cur = ws["mB"]
    a1 = f"{T}.layers.1.memory_segment_fusion_attention"
    errs["q"] = O.rel_l2(ws["q"], r(lin(cur, a1 + ".q_proj")))
    K1, V1 = kv[:, 2 * D:3 * D], kv[:, 3 * D:4 * D]
    ctx, lse2, col, _ = O.attention_heads(ws["q"], K1, V1, H, mode, want_colsum=True)
    errs["ctx"] = O.rel_l2(ws["ctx"], r(ctx))
    errs["lse2"] = float(np.abs(ws["lse2"] - lse2).max())
    pre_a = lin(ws["ctx"], a1 + ".residual.dense") + cur
    errs["a(LN)"] = O.rel_l2(ws["a"], r(O.layernorm(pre_a, w[a1 + ".residual.layernorm.weight"],
                                                  w[a1 + ".residual.layernorm.bias"], cfg.eps)))
    errs["h(relu)"] = O.rel_l2(ws["h"], r(np.maximum(lin(ws["a"], f"{T}.layers.1.mlp.0"), 0)))
    errs["pre(fp32)"] = O.rel_l2(ws["pre"], lin(ws["h"], f"{T}.layers.1.residual.dense"))   # dense + bias, fp32
    errs["m_out(LN)"] = O.rel_l2(to_np(cache[-1]).reshape(R, D),                              # + residual in the LN kernel
                                 r(O.layernorm(ws["pre"] + ws["a"], w[f"{T}.layers.1.residual.layernorm.weight"],
                                               w[f"{T}.layers.1.residual.layernorm.bias"], cfg.eps)))
    if score_mode == 0:
        part = eng.colsum_part(S // 196).cpu().numpy()
        errs["colsum"] = O.rel_l2(part, col)
    errs["scores"] = O.rel_l2(to_np(scores[-1]), r(col.sum(0).reshape(2, 196).mean(1)))
    print({k: f"{v:.1e}" for k, v in errs.items()})
    for k, v in errs.items():
        lim = {"pre(fp32)": 1e-5, "lse2": 1e-4, "scores": 4e-3}.get(k, TOL)
        assert v < lim, (k, v)
    # layer 0 as a whole (7 rounded stages): calibrated chain gate
    with O.accumulate_in(np.float32):
        l0, _ = O.transformer_layer(ws["mA"], x, w, f"{T}.layers.0", cfg, mode, False)
    with O.accumulate_in(np.float64):
        l0b, _ = O.transformer_layer(ws["mA"], x, w, f"{T}.layers.0", cfg, mode, False)
    assert O.rel_l2(cur, l0) < chain_tol(O.rel_l2(l0b, l0))


def test_fifo_eviction_vs_oracle():
    H, M, cap = 1, 2, 3
    cfg = O.PathConfig(hidden=128, heads=H, mem_tokens=M, depth=2, cache_cap=cap)
    w = O.make_weights(cfg, seed=6)
    proj = make_projector(cfg, w, cache_cap=cap)
    segs = [O.bf16_round(O.hash_normal_like((1 + t % 2, 196, 128), 950 + t)) for t in range(6)]
    ref = run_oracle_steps(cfg, w, "bf16", segs, np.float32)
    alt = run_oracle_steps(cfg, w, "bf16", segs, np.float64)
    proj.memory_cache = []
    with torch.no_grad():
        for t, seg in enumerate(segs):
            cache, _ = proj(to_dev(seg))
            ocache = ref[t][0]
            assert len(cache) == len(ocache) == min(t + 1, cap)
            for i in range(len(cache)):      # age order, oldest first
                floor = O.rel_l2(alt[t][0][i], ocache[i])
                assert O.rel_l2(to_np(cache[i]), ocache[i]) < chain_tol(floor), (t, i)
    # reset protocol: assigning [] restarts from the initial memory
    proj.memory_cache = []
    rm = O.RecurrentMemory(cfg, w, "bf16")
    rm.reset()
    with torch.no_grad():
        seg = O.bf16_round(O.hash_normal_like((2, 196, 128), 999))
        cache, _ = proj(to_dev(seg))
        ocache, _ = rm.step(seg)
    assert len(cache) == 1 and O.rel_l2(to_np(cache[0]), ocache[0]) < chain_tol(0.0) * 1.5
    with pytest.raises(capi.MavlmError):
        proj.memory_cache = [cache[0]]


@pytest.mark.parametrize("tag,M,F,steps", [("m8f32", 8, 32, 3), ("m64f8", 64, 8, 1), ("m64f32", 64, 32, 2)])
def test_golden_g7_reference_fullsize(tag, M, F, steps):
    """Against outputs of the REFERENCE ITSELF (its fp32 CPU run at D=1024, generated by tests/golden/make_golden.py):
    strided samples, norms, frame scores.  m64f32 is the exact bench.py workload (64 memory tokens, two 32-frame chunks:
    formation, then evolution + formation).  Gates that do not involve the oracle:
      * HIP-bf16 is at least as close to the reference's fp32 result as the reference's OWN bf16 run is (no slack),
      * the error has no bias: |mean signed error| is within 4 standard errors of zero,
      * the error does not grow faster than the reference's own bf16 error does from step to step."""
    z, m = load_golden("g7_fullsize.npz")
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    proj = make_projector(cfg, w)
    proj.memory_cache = []
    errs, ref_errs = [], []
    with torch.no_grad():
        for t in range(steps):
            seg = O.bf16_round(O.hash_normal_like((F, 196, 1024), m["segseed0"] + t))
            cache, scores = proj(to_dev(seg))
            mem = to_np(cache[-1]).reshape(-1)
            ref = z[f"{tag}_s{t}_sample"]
            got = mem[::m["stride"]]
            err = O.rel_l2(got, ref)
            err_refbf16 = O.rel_l2(z[f"{tag}_s{t}_sample_refbf16"], ref)
            d = (got - ref).astype(np.float64)
            bias, sem = d.mean(), d.std() / math.sqrt(d.size)
            print(f"{tag} step {t}: HIP-bf16 vs ref-fp32 {err:.2e}; reference-bf16 vs ref-fp32 {err_refbf16:.2e}; "
                  f"mean signed error {bias:+.2e} (standard error {sem:.1e})")
            assert err < TOL_REF_FP32
            assert err <= err_refbf16                         # inside the reference's own bf16 envelope, no slack
            assert abs(bias) < 4.0 * sem + 1e-6               # unbiased
            assert abs(np.linalg.norm(mem.astype(np.float64)) / float(z[f"{tag}_s{t}_norm"]) - 1) < 5e-3
            assert O.rel_l2(to_np(scores[-1]), z[f"{tag}_s{t}_scores"]) < 1e-2
            errs.append(err)
            ref_errs.append(err_refbf16)
    for t in range(1, steps):                                 # drift along the chain: no faster than the reference's own
        assert errs[t] / errs[t - 1] <= 1.25 * ref_errs[t] / ref_errs[t - 1] + 0.05, (errs, ref_errs)


@pytest.mark.parametrize("tag,M,F,steps", [("m8f32", 8, 32, 3), ("m64f32", 64, 32, 2)])
def test_golden_g7_reference_fullsize_fp16(tag, M, F, steps):
    """The fp16 path (MFMA f16, BASELINE.json configs[4]) against the REFERENCE's fp32 run at a flat tolerance: 16-bit floats
    with 10 mantissa bits sit inside the north-star's 1e-3 at the first steps (the reference's own fp16 run: 6.4e-4 -> 1.1e-3
    over four steps) - gate 1.5e-3 rel-L2 per step as the survey's protocol states, unbiased, frame scores
    5e-3.  (bf16 cannot meet a flat 1e-3 against fp32 - one rounding of an exact result costs 1.6e-3 - and is gated against the
    reference's own bf16 run instead, test_golden_g7_reference_fullsize.)"""
    z, m = load_golden("g7_fullsize.npz")
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    proj = make_projector(cfg, w, "fp16")
    proj.memory_cache = []
    with torch.no_grad():
        for t in range(steps):
            seg = O.bf16_round(O.hash_normal_like((F, 196, 1024), m["segseed0"] + t))      # (bf16-grid inputs are exact in fp16)
            cache, scores = proj(to_dev(seg, "fp16"))
            mem = to_np(cache[-1]).reshape(-1)
            ref = z[f"{tag}_s{t}_sample"]
            got = mem[::m["stride"]]
            err = O.rel_l2(got, ref)
            d = (got - ref).astype(np.float64)
            bias, sem = d.mean(), d.std() / math.sqrt(d.size)
            print(f"{tag} fp16 step {t}: HIP-fp16 vs ref-fp32 {err:.2e}; mean signed error {bias:+.2e} (standard error {sem:.1e})")
            assert err < 1.5e-3
            assert abs(bias) < 4.0 * sem + 1e-6
            assert O.rel_l2(to_np(scores[-1]), z[f"{tag}_s{t}_scores"]) < 5e-3


def _g7_gates(got, mem, scores, z, pfx, t, m, label):
    """the gates of the G7 tests for one step of one video; returns (err, reference's own bf16 err)"""
    ref = z[f"{pfx}s{t}_sample"]
    err, err_refbf16 = O.rel_l2(got, ref), O.rel_l2(z[f"{pfx}s{t}_sample_refbf16"], ref)
    d = (got - ref).astype(np.float64)
    bias, sem = d.mean(), d.std() / math.sqrt(d.size)
    print(f"{label} step {t}: HIP-bf16 vs ref-fp32 {err:.2e}; reference-bf16 vs ref-fp32 {err_refbf16:.2e}; "
          f"mean signed error {bias:+.2e} (standard error {sem:.1e})")
    assert err < TOL_REF_FP32 and err <= err_refbf16          # inside the reference's own bf16 envelope, no slack
    assert abs(bias) < 4.0 * sem + 1e-6                       # unbiased
    assert abs(np.linalg.norm(mem.astype(np.float64)) / float(z[f"{pfx}s{t}_norm"]) - 1) < 5e-3
    assert O.rel_l2(scores, z[f"{pfx}s{t}_scores"]) < 1e-2
    return err, err_refbf16


def test_golden_g7_wide_reference_fullsize():
    """The OneVision-7B width against the REFERENCE (round 4; tests/golden/g7_wide_fullsize.npz: hidden 3584, head_dim 448, 8
    memory tokens, 3 steps of 2 / 1 / 2 frames of the imported reference, fp32 + its own bf16 run): the wide-head kernels
    (attention_hd.hip, 32-query waves / split-KV at this row count, the frame scores riding on the forward's tile entries) and the K = 3584 /
    14336 GEMMs through the G7 gates - at least as close to the reference's fp32 result as its own bf16 run, unbiased, no
    faster drift.  Then the same video twice through a ROW BATCH of two (`BatchedProjector`): every video passes the same
    gates (the batch runs the wide-head attention per video and the stacked GEMMs / LayerNorms)."""
    from memory_augmented_vlm_amd.model.memory_module.MemoryController import BatchedProjector
    z, m = load_golden("g7_wide_fullsize.npz")
    cfg = O.PathConfig(hidden=3584, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    proj = make_projector(cfg, w)
    segs = [to_dev(O.bf16_round(O.hash_normal_like((F, 196, 3584), m["segseed0"] + t))) for t, F in enumerate(m["frames"])]
    proj.memory_cache = []
    errs, ref_errs = [], []
    with torch.no_grad():
        for t, seg in enumerate(segs):
            cache, scores = proj(seg)
            mem = to_np(cache[-1]).reshape(-1)
            e, r = _g7_gates(mem[::m["stride"]], mem, to_np(scores[-1]), z, "", t, m, "7B width")
            errs.append(e)
            ref_errs.append(r)
        for t in range(1, len(errs)):
            assert errs[t] / errs[t - 1] <= 1.25 * ref_errs[t] / ref_errs[t - 1] + 0.05, (errs, ref_errs)
        bp = BatchedProjector(proj, 2)
        bp.reset()
        for t, seg in enumerate(segs):
            sc = bp.step([seg, seg.clone()])
            for b in range(2):
                mem = to_np(bp.memory_cache(b)[-1]).reshape(-1)
                _g7_gates(mem[::m["stride"]], mem, to_np(sc[b]), z, "", t, m, f"7B width, row batch video {b}")


def test_wide_head_frame_scores_ride_on_the_forward():
    """head_dim 448 (round 4): the frame scores come out of the last formation layer's forward (`attn_fwd_hd2_kernel<.., FT = 1>` writes
    one log-mass entry per query row and 32-key tile, `frame_tiles_kernel` + `frame_finish_kernel` add them up) instead of the
    column-sum pass that recomputed Q.K^T.  Against that pass (`mavlm_set_frame_score_mode(0)`) on the same inputs: the memory is
    bit-identical (asking for scores, or how, never changes it), the scores agree to the rounding of their 16-bit storage; full-size
    chunk (32 frames: R = 1568, S = 6272 - the split-KV schedule of a single video), a ragged one (5 frames: the last 32-key tile is
    cut by S and frame boundaries fall at every offset), and a row batch of two (the grid's z dimension)."""
    from memory_augmented_vlm_amd.model.memory_module.MemoryController import BatchedProjector
    cfg = O.PathConfig(hidden=3584, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=77)
    proj = make_projector(cfg, w)
    segs = [to_dev(O.bf16_round(O.hash_normal_like((F, 196, 3584), 7700 + t))) for t, F in enumerate((32, 5, 32))]
    lib = capi.lib()

    def run(mode):
        capi.check(lib.mavlm_set_frame_score_mode(mode), "mode")
        try:
            proj.memory_cache = []
            out = []
            with torch.no_grad():
                for seg in segs:
                    cache, scores = proj(seg)
                    out.append((cache[-1].clone(), scores[-1].clone()))
                bp = BatchedProjector(proj, 2)
                bp.reset()
                for seg in segs[:2]:
                    sc = bp.step([seg, segs[0][:seg.shape[0]].clone()])
                    out.append((bp.memory_cache(1)[-1].clone(), sc[1].clone()))
            return out
        finally:
            lib.mavlm_set_frame_score_mode(1)
    fused, passes = run(1), run(0)
    for (m1, s1), (m0, s0) in zip(fused, passes):
        assert torch.equal(m1, m0)
        assert s1.shape == s0.shape and O.rel_l2(to_np(s1), to_np(s0)) < 5e-3
        assert abs(float(s1.float().sum()) - 8 * 8) < 2e-2 * 64          # sum_f score_f = H * R / P (MemoryController.py:135-139)


@pytest.mark.parametrize("M,frames", [(8, (32, 5, 3)), (24, (9, 4))])
def test_frame_scores_from_tile_entries_on_small_grids(M, frames):
    """head_dim 128, round 4: the small grids that split their keys (the checkpoint's 8 memory tokens, one video) had kept the
    column-sum pass - the per-(row, frame) masses of `attn_fwd3_kernel<.., FR = 1>` need one writer per entry.  They now write one
    log-mass entry per (row, 64-key tile) (`FR = 2`, any schedule) and `frame_tiles_kernel` adds them per frame.  Mode 1 (automatic:
    tile entries where the keys are split, per-frame masses elsewhere), mode 2 (tile entries wherever supported) and mode 0 (the
    column-sum pass) on the same chunks: the memory is bit-identical, the scores agree to the rounding of their 16-bit storage."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=78)
    proj = make_projector(cfg, w)
    segs = [to_dev(O.bf16_round(O.hash_normal_like((F, 196, 1024), 7800 + t))) for t, F in enumerate(frames)]
    lib = capi.lib()

    def run(mode):
        capi.check(lib.mavlm_set_frame_score_mode(mode), "mode")
        try:
            proj.memory_cache = []
            out = []
            with torch.no_grad():
                for seg in segs:
                    cache, scores = proj(seg)
                    out.append((cache[-1].clone(), scores[-1].clone()))
            return out
        finally:
            lib.mavlm_set_frame_score_mode(1)
    ref = run(0)
    for mode in (1, 2):
        for (m1, s1), (m0, s0) in zip(run(mode), ref):
            assert torch.equal(m1, m0)
            assert s1.shape == s0.shape and O.rel_l2(to_np(s1), to_np(s0)) < 5e-3
            assert abs(float(s1.float().sum()) - 8 * M) < 2e-2 * 8 * M
    # a row batch of two under the forced tile-entry form (the (video, head) pairs of a batch share one entry array)
    from memory_augmented_vlm_amd.model.memory_module.MemoryController import BatchedProjector

    def run_batch(mode):
        capi.check(lib.mavlm_set_frame_score_mode(mode), "mode")
        try:
            bp = BatchedProjector(proj, 2)
            bp.reset()
            out = []
            with torch.no_grad():
                for seg in segs:
                    sc = bp.step([seg, segs[0][:seg.shape[0]].clone()])
                    out.append([(bp.memory_cache(b)[-1].clone(), sc[b].clone()) for b in range(2)])
            return out
        finally:
            lib.mavlm_set_frame_score_mode(1)
    for a, b in zip(run_batch(2), run_batch(0)):
        for (m2, s2), (m0, s0) in zip(a, b):
            assert torch.equal(m2, m0)
            assert O.rel_l2(to_np(s2), to_np(s0)) < 5e-3


@pytest.mark.parametrize("hidden,hd", [(1024, 128), (3584, 448)])
def test_splitk_reduction_inside_the_layernorm_kernel_same_bits(hidden, hd):
    """Round 4: where the 4D -> D projection splits its contraction (few memory tokens), the fp32 planes go straight into one reduce
    + bias + residual + LayerNorm kernel (`mavlm_launch_layernorm_planes`) instead of a reduction pass, an fp32 dense output and the
    LayerNorm kernel (`mavlm_set_splitk_layernorm(0)`): the same additions in the same order - memories and scores bit for bit."""
    cfg = O.PathConfig(hidden=hidden, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=79)
    proj = make_projector(cfg, w)
    segs = [to_dev(O.bf16_round(O.hash_normal_like((F, 196, hidden), 7900 + t))) for t, F in enumerate((32, 7, 2))]
    lib = capi.lib()

    def run(on):
        capi.check(lib.mavlm_set_splitk_layernorm(on), "splitk layernorm")
        try:
            proj.memory_cache = []
            out = []
            with torch.no_grad():
                for seg in segs:
                    cache, scores = proj(seg)
                    out.append((cache[-1].clone(), scores[-1].clone()))
            return out
        finally:
            lib.mavlm_set_splitk_layernorm(1)
    for (m1, s1), (m0, s0) in zip(run(1), run(0)):
        assert torch.equal(m1, m0) and torch.equal(s1, s0)


def test_golden_g7_fifo_wrap_fullsize():
    """The reference-pinned chain PAST the FIFO's capacity at full width (round 3): checkpoint shape (8 memory tokens,
    D = 1024), 13 steps of 1-2 frames, cap 10 - eviction at steps 10-12 (MemoryController.py:152-154), the evolution attends
    over all 10 cached memories.  HIP bf16 against the REFERENCE's fp32 run with the gates of the G7 test (no oracle
    involved): at least as close as the reference's own bf16 run, unbiased, no faster drift; plus the surviving FIFO
    entries after the last step (ring order -> FIFO order)."""
    z, m = load_golden("g7_fifo_fullsize.npz")
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    proj = make_projector(cfg, w)
    proj.memory_cache = []
    errs, ref_errs = [], []
    for t, F in enumerate(m["frames"]):
        seg = O.bf16_round(O.hash_normal_like((F, 196, 1024), m["segseed0"] + t))
        cache, scores = proj(to_dev(seg))
        assert len(cache) == min(t + 1, 10)
        mem = to_np(cache[-1]).reshape(-1)
        ref = z[f"s{t}_sample"]
        got = mem[::m["stride"]]
        err, err_refbf16 = O.rel_l2(got, ref), O.rel_l2(z[f"s{t}_sample_refbf16"], ref)
        d = (got - ref).astype(np.float64)
        bias, sem = d.mean(), d.std() / math.sqrt(d.size)
        print(f"fifo step {t} ({F} frames): HIP-bf16 vs ref-fp32 {err:.2e}; reference-bf16 vs ref-fp32 {err_refbf16:.2e}; "
              f"mean signed error {bias:+.2e} (standard error {sem:.1e})")
        assert err < TOL_REF_FP32 and err <= err_refbf16
        assert abs(bias) < 4.0 * sem + 1e-6
        assert abs(np.linalg.norm(mem.astype(np.float64)) / float(z[f"s{t}_norm"]) - 1) < 5e-3
        assert O.rel_l2(to_np(scores[-1]), z[f"s{t}_scores"]) < 1e-2
        errs.append(err)
        ref_errs.append(err_refbf16)
    for t in range(1, len(errs)):
        assert errs[t] / errs[t - 1] <= 1.25 * ref_errs[t] / ref_errs[t - 1] + 0.05, (errs, ref_errs)
    final = np.stack([to_np(c).reshape(-1)[::m["stride"]] for c in cache])
    assert final.shape == z["final_cache_samples"].shape
    for i in range(final.shape[0]):                           # oldest first: steps 3..12
        assert O.rel_l2(final[i], z["final_cache_samples"][i]) < TOL_REF_FP32, i


def _tiny_host(cfg: O.PathConfig, w, mode="bf16", vocab=48900):
    # LlavaMetaModel calls super().__init__(config): give it a base that accepts it
    class Base(torch.nn.Module):
        def __init__(self, config):
            super().__init__()
            self.embed_tokens = torch.nn.Embedding(vocab, config.hidden_size)

    class Model(arch.LlavaMetaModel, Base):
        pass

    hf = types.SimpleNamespace(hidden_size=cfg.hidden, num_memory_tokens=cfg.mem_tokens, mm_patch_merge_type="spatial_unpad",
                               mm_newline_position="one_token", mm_spatial_pool_mode="bilinear",
                               tokenizer_model_max_length=32768, tokenizer_padding_side="right")
    model = Model(hf).eval()
    model.image_newline = torch.nn.Parameter(torch.zeros(cfg.hidden))
    load_oracle_weights(model, w)
    return model.to("cuda").to(DT[mode]), hf


def test_video_tokens_vs_oracle():
    """PE add -> 2 chunks (32 + 8 frames, evolution) -> fuser + type add + concat, whole block vs the oracle."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=8)
    model, _ = _tiny_host(cfg, w)
    rows = sorted(set(O.MEM_PROMPT_IDS + O.FRAME_PROMPT_IDS))
    emb = np.zeros((48900, 1024), np.float32)
    emb[rows] = O.bf16_round(O.hash_normal_like((len(rows), 1024), 81, 0.02))
    with torch.no_grad():
        model.embed_tokens.weight.copy_(to_dev(emb))
    T = 40
    x = O.bf16_round(O.hash_normal_like((T, 196, 1024), 82))
    idx = O.subsample_indices(45)[:T]
    toks, parts = O.video_tokens(x, idx, cfg, w, emb, "bf16", return_parts=True)
    with O.accumulate_in(np.float64):
        toks64 = O.video_tokens(x, idx, cfg, w, emb, "bf16")
    dev = "cuda"
    mp = model.embed_tokens(torch.tensor(O.MEM_PROMPT_IDS, device=dev))
    fp = model.embed_tokens(torch.tensor(O.FRAME_PROMPT_IDS, device=dev))
    got, info = arch.video_memory_tokens(model, to_dev(x), torch.from_numpy(idx), mp, fp, model.image_newline)
    assert got.shape[0] == toks.shape[0] == 10 + 2 * 1568 + 1 + 9 + 32 * 196 + 1
    g = to_np(got)
    assert O.rel_l2(to_np(info["pe_frames"]), parts["pe"]) < 1e-6
    a, b = info["memory_rows"]
    floor = O.rel_l2(toks64[a:b], toks[a:b])
    err = O.rel_l2(g[a:b], toks[a:b])
    print(f"fused memory tokens: HIP vs oracle {err:.2e}, noise floor {floor:.2e}")
    assert err < chain_tol(floor)
    np.testing.assert_array_equal(g[:a], toks[:a])                         # prompt rows are copies
    assert O.rel_l2(g[b:], toks[b:]) < 1e-6                                # newline, prompt, fine frames (+E1)
    # frame-dropout branch: memory half only
    got2, _ = arch.video_memory_tokens(model, to_dev(x), torch.from_numpy(idx), mp, fp, model.image_newline,
                                       with_frames=False)
    assert got2.shape[0] == b + 1 and torch.equal(got2, got[:b + 1])       # deterministic + same prefix
    with pytest.raises(ValueError, match="exceed max_frames"):
        arch.video_memory_tokens(model, to_dev(x[:2]), torch.tensor([0, 600]), mp, fp, model.image_newline)


def test_prepare_inputs_end_to_end_vs_oracle():
    """The reference-shaped outer API on a toy host: fake vision tower -> 2x2 bilinear pool -> memory path ->
    splice into the text sequence; compared against the oracle pipeline fed the same features."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=9)
    model, hf = _tiny_host(cfg, w)
    side, F0, D = 27, 40, 1024
    rows = sorted(set(O.MEM_PROMPT_IDS + O.FRAME_PROMPT_IDS + [11, 22, 33, 44]))
    emb = np.zeros((48900, D), np.float32)
    emb[rows] = O.bf16_round(O.hash_normal_like((len(rows), D), 91, 0.02))
    feats = O.bf16_round(O.hash_normal_like((F0, side * side, D), 92))
    with torch.no_grad():
        model.embed_tokens.weight.copy_(to_dev(emb))
    table = to_dev(feats)

    class Tower(torch.nn.Module):
        num_patches_per_side = side

        def forward(self, images):
            return table[images.reshape(-1).long()]

    model.vision_tower = Tower()
    model.mm_projector = torch.nn.Identity()

    class LM(arch.LlavaMetaForCausalLM, torch.nn.Module):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.config = hf
            self.model = model

        def get_model(self):
            return self.model

        @property
        def device(self):
            return torch.device("cuda")

    lm = LM().eval()
    ids = torch.tensor([[11, 22, arch.IMAGE_TOKEN_INDEX, 33, 44]], device="cuda")
    labels = torch.tensor([[-100, -100, -100, 33, 44]], device="cuda")
    am = torch.ones_like(ids)
    images = [torch.arange(F0, dtype=torch.float32, device="cuda").reshape(F0, 1, 1, 1)]
    with torch.no_grad():
        out = lm.prepare_inputs_labels_for_multimodal(ids, None, am, None, labels, images, modalities=["video"])
    none_ids, pos, mask, pkv, embeds, labs = out
    assert none_ids is None and pos is None and pkv is None
    # oracle pipeline (bf16 emulation); the pooled features come from the GPU pool so that only the path is compared
    idx = O.subsample_indices(F0)
    assert len(idx) == 64
    pooled = to_np(lm.get_2dPool(table[torch.from_numpy(idx).cuda()]))
    assert O.rel_l2(pooled, O.bf16_round(O.bilinear_pool(feats[idx], side))) < 3e-3     # torch bf16 interpolate
    toks = O.video_tokens(pooled, idx, cfg, w, emb, "bf16")
    with O.accumulate_in(np.float64):
        toks64 = O.video_tokens(pooled, idx, cfg, w, emb, "bf16")
    e, lab, msk, pid = O.splice(to_np(ids).astype(np.int64), to_np(labels).astype(np.int64), np.ones((1, 5)), toks,
                                O.bf16_round(emb), max_len=32768)
    assert tuple(embeds.shape) == e.shape
    assert O.rel_l2(to_np(embeds)[0, 2:-2], e[0, 2:-2]) < chain_tol(O.rel_l2(toks64, toks))
    # the in-place emit used above (batch 1, one placeholder) equals the general splice path bit for bit
    with torch.no_grad():
        tokens2, _ = arch.video_memory_tokens(model, lm.get_2dPool(table[torch.from_numpy(idx).cuda()]),
                                              torch.from_numpy(idx), model.embed_tokens(torch.tensor(O.MEM_PROMPT_IDS, device="cuda")),
                                              model.embed_tokens(torch.tensor(O.FRAME_PROMPT_IDS, device="cuda")),
                                              model.image_newline)
        gen = arch.splice_into_text(lm, model, [tokens2], ids, None, am, None, labels)
    assert torch.equal(gen[4], embeds) and torch.equal(gen[5], labs) and torch.equal(gen[2], mask)
    assert arch.video_token_rows(64, 8) == tokens2.shape[0]
    np.testing.assert_array_equal(labs.cpu().numpy(), lab)
    assert bool(mask.all())
    # hipGraph replay behind the reference's entry point (round 4, `enable_memory_graphs`): the first occurrence of a video
    # shape runs eagerly, the second captures the whole launch sequence on the model's own engine, later ones replay it -
    # all bit-identical to the eager result, for changing frame CONTENTS (the graph reads its static input copy)
    lm.enable_memory_graphs(2)
    with torch.no_grad():
        for rep in range(4):
            table.copy_(to_dev(O.bf16_round(O.hash_normal_like((F0, side * side, D), 920 + rep))))
            rm_ = model.recurrent_memory_transformer
            lm._mem_graph_capacity = 0
            want = lm.prepare_inputs_labels_for_multimodal(ids, None, am, None, labels, images, modalities=["video"])[4].clone()
            want_cache = [c_.clone() for c_ in rm_.memory_cache]
            rm_.memory_cache = []                               # (host-side state of another video in between)
            lm._mem_graph_capacity = 2
            got = lm.prepare_inputs_labels_for_multimodal(ids, None, am, None, labels, images, modalities=["video"])[4]
            assert torch.equal(got, want), rep
            # the module's host-side cache after a replayed forward is the replayed video's (ring views, oldest first)
            assert len(rm_.memory_cache) == len(want_cache) == 2
            assert all(torch.equal(a_, b_) for a_, b_ in zip(rm_.memory_cache, want_cache))
            assert len(lm._mem_graphs) == (1 if rep >= 1 else 0)
        # a second shape gets its own graph; the eager path of the first shape still works next to them
        images2 = [torch.arange(33, dtype=torch.float32, device="cuda").reshape(33, 1, 1, 1)]
        outs2 = [lm.prepare_inputs_labels_for_multimodal(ids, None, am, None, labels, images2, modalities=["video"])[4].clone()
                 for _ in range(3)]
        assert torch.equal(outs2[0], outs2[1]) and torch.equal(outs2[1], outs2[2]) and len(lm._mem_graphs) == 2
        lm._mem_graph_capacity = 0
        assert torch.equal(lm.prepare_inputs_labels_for_multimodal(ids, None, am, None, labels, images, modalities=["video"])[4], want)


def test_baseline_size_properties():
    """BASELINE config 2 sizes (64 frames, 64 memory tokens, D=1024): properties that need no CPU oracle run."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=64, depth=2)
    w = O.make_weights(cfg, seed=10)
    model, _ = _tiny_host(cfg, w)
    x = to_dev(O.bf16_round(O.hash_normal_like((64, 196, 1024), 101)))
    idx = torch.arange(64)
    mp = torch.zeros((10, 1024), device="cuda", dtype=torch.bfloat16)
    fp = torch.ones((9, 1024), device="cuda", dtype=torch.bfloat16)
    a, info = arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
    b, _ = arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
    assert torch.equal(a, b)                                   # idempotent / deterministic, state fully reset
    R = 64 * 196
    assert a.shape[0] == 10 + 2 * R + 1 + 9 + 32 * 196 + 1
    assert torch.isfinite(a.float()).all()
    rm = model.recurrent_memory_transformer
    for s in rm.frame_attn_scores[-2:]:
        assert abs(float(s.float().sum()) - 8 * 64) < 0.02 * 8 * 64          # sum_f score_f = H*R/P
    mem = torch.stack(list(rm.memory_cache)).float()           # LayerNorm output: per-row mean/var of (y-beta)/gamma
    g = rm.layers[1].residual.layernorm.weight.float()
    bb = rm.layers[1].residual.layernorm.bias.float()
    z = (mem - bb) / g
    assert float(z.mean(-1).abs().max().detach()) < 2e-2
    assert float((z.var(-1, unbiased=False) - 1).abs().max().detach()) < 5e-2


def test_pool_two_videos_in_flight_bit_identical():
    """MemoryPathPool: two videos on two streams over shared weights == the serial path, bit for bit."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=13)
    model, _ = _tiny_host(cfg, w)
    vids = [(to_dev(O.bf16_round(O.hash_normal_like((40 if i % 2 else 33, 196, 1024), 1300 + i))),
             torch.arange(40 if i % 2 else 33)) for i in range(4)]
    mp = torch.randn((10, 1024), device="cuda").bfloat16()
    fp = torch.randn((9, 1024), device="cuda").bfloat16()
    serial = [arch.video_memory_tokens(model, v, i, mp, fp, model.image_newline)[0].clone() for v, i in vids]
    pool = arch.MemoryPathPool(model, 2)
    assert pool.slots[1].recurrent_memory_transformer.layers[0].mlp[0].weight is \
        model.recurrent_memory_transformer.layers[0].mlp[0].weight          # shared parameters, no copy
    outs = pool.run(vids, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    for a, b in zip(serial, outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M", [8, 64])
def test_graph_capture_replay_bit_identical(M):
    """The whole per-video launch sequence captured into one hipGraph replays to the same bits as eager launches - at the
    checkpoint's 8 memory tokens and at the metric's 64, where the Residual blocks run as the fused kernel whose workgroups
    exchange row statistics through memory (launch-counter epochs: nothing is re-zeroed between replays) and the attention
    runs its stream-K schedule."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=14)
    model, _ = _tiny_host(cfg, w)
    T = 70
    idx = torch.arange(T)
    mp = torch.randn((10, 1024), device="cuda").bfloat16()
    fp = torch.randn((9, 1024), device="cuda").bfloat16()
    g = arch.GraphedVideoMemory(model, T, idx)
    for seed in (1400, 1401):
        x = to_dev(O.bf16_round(O.hash_normal_like((T, 196, 1024), seed)))
        eager, _ = arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
        out = g(x, mp, fp, model.image_newline)
        torch.cuda.synchronize()
        assert out.shape == eager.shape == (10 + 3 * M * 196 + 1 + 9 + 32 * 196 + 1, 1024)
        assert torch.equal(out, eager)
    for eng in (model.recurrent_memory_transformer._engine, g.view.recurrent_memory_transformer._engine):
        st = eng.ln_exchange_status() if eng is not None else None
        if M == 64:
            assert st is not None and st[0] > 0           # the fused Residual kernel did run (launch counter advanced)
        if st is not None:
            assert st[1] == 0                             # no workgroup ever gave up waiting for a partner


@pytest.mark.parametrize("T", [1, 5, 33])
def test_short_and_ragged_videos_vs_oracle(T):
    """Edge cases of the chunking (llava_arch.py:437-457, segment.py:169-192): a single frame, a short single chunk,
    and 32 + 1 frames (a one-frame tail chunk that still goes through memory evolution)."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=15)
    model, _ = _tiny_host(cfg, w)
    emb = np.zeros((48900, 1024), np.float32)
    x = O.bf16_round(O.hash_normal_like((T, 196, 1024), 1500 + T))
    idx = np.arange(T) * 3 % 600
    toks = O.video_tokens(x, idx, cfg, w, emb, "bf16")
    with O.accumulate_in(np.float64):
        toks64 = O.video_tokens(x, idx, cfg, w, emb, "bf16")
    mp = torch.zeros((10, 1024), device="cuda", dtype=torch.bfloat16)
    fp = torch.zeros((9, 1024), device="cuda", dtype=torch.bfloat16)
    got, info = arch.video_memory_tokens(model, to_dev(x), torch.from_numpy(idx), mp, fp, model.image_newline)
    n = 1 if T <= 32 else 2
    assert info["num_memories"] == n and got.shape[0] == toks.shape[0] == 10 + n * 1568 + 1 + 9 + min(32, T) * 196 + 1
    assert O.rel_l2(to_np(got), toks) < chain_tol(O.rel_l2(toks64, toks))


def test_fifo_saturation_full_width_vs_oracle():
    """12 one-frame chunks at D=1024, M=8: the FIFO (cap 10) saturates, the evolution attends over 10 x 1568 keys and
    the ring wraps (MemoryController.py:152-154).  Oldest-first order of the returned cache is checked too."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=16)
    proj = make_projector(cfg, w)
    segs = [O.bf16_round(O.hash_normal_like((1, 196, 1024), 1600 + t)) for t in range(12)]
    ref = run_oracle_steps(cfg, w, "bf16", segs, np.float32)
    alt = run_oracle_steps(cfg, w, "bf16", segs, np.float64)
    proj.memory_cache = []
    with torch.no_grad():
        for seg in segs:
            cache, scores = proj(to_dev(seg))
    assert len(cache) == len(ref[-1][0]) == 10
    for i in (0, 4, 9):
        floor = O.rel_l2(alt[-1][0][i], ref[-1][0][i])
        assert O.rel_l2(to_np(cache[i]), ref[-1][0][i]) < chain_tol(floor), i
    assert len(proj.frame_attn_scores) == 12


def test_config5_shape_fp16_128_memory_tokens_properties():
    """BASELINE configs[4] shape on one GPU: 128 memory tokens, fp16 MFMA path.  Too large for the CPU oracle in a
    test, so size-independent properties only."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=128, depth=2)
    w = O.make_weights(cfg, seed=17, grid="fp16")
    proj = make_projector(cfg, w, "fp16")
    proj.memory_cache = []
    with torch.no_grad():
        for t in range(3):
            seg = to_dev(O.fp16_round(O.hash_normal_like((32, 196, 1024), 1700 + t)), "fp16")
            cache, scores = proj(seg)
    assert len(cache) == 3 and tuple(cache[-1].shape) == (128, 196, 1024) and cache[-1].dtype == torch.float16
    assert torch.isfinite(torch.stack(list(cache)).float()).all()
    for s_ in scores[-3:]:
        assert abs(float(s_.float().sum()) - 8 * 128) < 0.02 * 8 * 128      # sum_f score_f = H*R/P


@pytest.mark.gpu
def test_vision_projector_mlp2x_gelu_matches_oracle():
    """§8f rank 1, first half: mm_projector (builder.py:41-48) = Linear(1152,D) GELU Linear(D,D) on the HIP GEMMs,
    state-dict keys of the reference's nn.Sequential; checked against the oracle's linear/gelu with bf16 rounding."""
    import types
    from memory_augmented_vlm_amd.model.multimodal_projector import build_vision_projector
    torch.manual_seed(5)
    cfg = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=1152, hidden_size=1024)
    proj = build_vision_projector(cfg).cuda().to(torch.bfloat16)
    assert sorted(proj.state_dict()) == ["0.bias", "0.weight", "2.bias", "2.weight"]
    x = (torch.randn(3, 729, 1152, device="cuda") * 0.5).to(torch.bfloat16)
    with torch.no_grad():
        y = proj(x)
    assert tuple(y.shape) == (3, 729, 1024)
    r = O.rounder("bf16")
    u = r(O.gelu_erf(O.linear(to_np(x).reshape(-1, 1152), to_np(proj[0].weight), to_np(proj[0].bias))))
    ref = r(O.linear(u, to_np(proj[2].weight), to_np(proj[2].bias))).reshape(3, 729, 1024)
    assert O.rel_l2(to_np(y), ref) < 1e-3
    with pytest.raises(NotImplementedError):
        build_vision_projector(types.SimpleNamespace(mm_projector_type="pooler", mm_hidden_size=8, hidden_size=8))
    with pytest.raises(ValueError):
        build_vision_projector(types.SimpleNamespace(mm_projector_type="bogus", mm_hidden_size=8, hidden_size=8))


def test_config3_shape_long_video_ov7b_hipgraph():
    """BASELINE configs[2] shape on one GPU: a 256-frame video in 32-frame chunks at the LLaVA-OneVision-7B width
    (D = 3584, 8 heads of 448 -> wide-head kernels), the whole per-video update captured in ONE hipGraph.
    Too large for the CPU oracle in a test: graph replay == eager launches bit for bit, FIFO of 8 memories, finite
    tokens, and the frame-score identity sum_f score_f = H*R/P on every chunk."""
    D, T = 3584, 256
    cfg = O.PathConfig(hidden=D, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=19)
    model, _ = _tiny_host(cfg, w)
    idx = torch.arange(T)
    mp = (torch.randn((10, D), device="cuda") * 0.02).bfloat16()
    fp = (torch.randn((9, D), device="cuda") * 0.02).bfloat16()
    x = to_dev(O.bf16_round(O.hash_normal_like((T, 196, D), 1900)))
    rm = model.recurrent_memory_transformer
    n_scores = len(rm.frame_attn_scores)
    eager, info = arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
    assert info["num_memories"] == 8 and len(rm.memory_cache) == 8
    assert eager.shape == (arch.video_token_rows(T, 8), D) and torch.isfinite(eager.float()).all()
    for s_ in rm.frame_attn_scores[n_scores:]:
        assert s_.shape == (32,) and abs(float(s_.float().sum()) - 8 * 8) < 0.02 * 64
    g = arch.GraphedVideoMemory(model, T, idx)
    out = g(x, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("depth,cap,H,hd,M,frames", [(1, 10, 8, 128, 8, [2, 1, 2]), (3, 1, 4, 128, 2, [1, 2, 1]),
                                                     (2, 2, 16, 64, 3, [1, 1, 1, 1]), (4, 10, 1, 128, 1, [1, 2])])
def test_config_corners_vs_oracle(depth, cap, H, hd, M, frames):
    """Corners of the configuration space: depth 1 (the reference Config default) up to the ABI maximum, a FIFO of one
    entry (every step evicts), 16 narrow heads, a single memory token; inference and training forward, bit-identical
    to each other and inside the chain tolerance of the oracle."""
    cfg = O.PathConfig(hidden=H * hd, heads=H, mem_tokens=M, depth=depth, cache_cap=cap)
    w = O.make_weights(cfg, seed=40 + depth)
    proj = make_projector(cfg, w, "bf16", cache_cap=cap)
    segs = [O.bf16_round(O.hash_normal_like((f, 196, cfg.hidden), 4000 + t)) for t, f in enumerate(frames)]
    ref = run_oracle_steps(cfg, w, "bf16", segs, np.float32)
    alt = run_oracle_steps(cfg, w, "bf16", segs, np.float64)
    proj.memory_cache = []
    for t, s_ in enumerate(segs):
        cache, scores = proj(to_dev(s_))
        assert len(cache) == min(t + 1, cap)
        floor = O.rel_l2(alt[t][0][-1], ref[t][0][-1])
        assert O.rel_l2(to_np(cache[-1]), ref[t][0][-1]) < chain_tol(floor)
        assert O.rel_l2(to_np(scores[-1]), ref[t][1]) < max(5e-3, 4 * O.rel_l2(alt[t][1], ref[t][1]))
    infer = [c.clone() for c in cache]
    proj.memory_cache = []
    with torch.enable_grad():
        for s_ in segs:
            cache, _ = proj(to_dev(s_))
        assert all(c.requires_grad for c in cache)
        for a, b in zip(cache, infer):
            assert torch.equal(a.detach(), b)
        sum((c.float() ** 2).mean() for c in cache).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad.float()).all() for n, p in proj.named_parameters()
               if not n.startswith("memory_update_attention"))
    proj.memory_cache = []


def test_options_no_frame_scores_and_oversized_chunk():
    """compute_frame_scores=False computes no scores (none appended) and is free of side effects: the memory is BIT-identical
    to the scored run (round 3: the launch that carries the frame masses runs the schedule of the plain forward); a chunk
    longer than the default 32 frames re-creates the engine with a larger workspace (first step only) and matches the
    oracle; changing the chunk size upward in the middle of a video is refused."""
    cfg = O.PathConfig(hidden=256, heads=2, mem_tokens=4, depth=2)
    w = O.make_weights(cfg, seed=51)
    proj = make_projector(cfg, w, "bf16")
    seg40 = O.bf16_round(O.hash_normal_like((40, 196, 256), 5100))
    seg3 = O.bf16_round(O.hash_normal_like((3, 196, 256), 5101))
    proj.memory_cache = []
    n0 = len(proj.frame_attn_scores)
    cache, scores = proj(to_dev(seg40))
    with_scores = cache[-1].clone()
    assert len(scores) == n0 + 1 and scores[-1].shape == (40,)
    ref = run_oracle_steps(cfg, w, "bf16", [seg40], np.float32)
    alt = run_oracle_steps(cfg, w, "bf16", [seg40], np.float64)
    assert O.rel_l2(to_np(with_scores), ref[0][0][-1]) < chain_tol(O.rel_l2(alt[0][0][-1], ref[0][0][-1]))
    proj.compute_frame_scores = False
    proj.memory_cache = []
    cache, scores = proj(to_dev(seg40))
    assert len(scores) == n0 + 1
    assert torch.equal(cache[-1], with_scores)
    capi.check(capi.lib().mavlm_set_frame_score_mode(0), "mode")                     # column-sum pass: the plan does not change
    try:
        proj.compute_frame_scores = True
        proj.memory_cache = []
        scored0 = proj(to_dev(seg40))[0][-1].clone()
        proj.compute_frame_scores = False
        proj.memory_cache = []
        assert torch.equal(proj(to_dev(seg40))[0][-1], scored0)
    finally:
        capi.lib().mavlm_set_frame_score_mode(1)
    proj(to_dev(seg3))                                   # smaller chunks are fine
    small = make_projector(cfg, w, "bf16")
    small.memory_cache = []
    small(to_dev(seg3))
    with pytest.raises(capi.MavlmError, match="middle of a video"):
        small(to_dev(seg40))


def test_long_video_1024_frames():
    """BASELINE configs[3] frame count on one GPU: 1024 frames = 32 chunks, the FIFO wraps three times.  Small width
    against the oracle (every step of the chain inside the calibrated tolerance), full width (D = 1024, M = 8) as
    properties: eager == hipGraph bit for bit, 10 memories kept, finite tokens."""
    cfg = O.PathConfig(hidden=256, heads=2, mem_tokens=2, depth=2)
    w = O.make_weights(cfg, seed=61)
    proj = make_projector(cfg, w, "bf16")
    segs = [O.bf16_round(O.hash_normal_like((32, 196, 256), 6100 + t)) for t in range(32)]
    ref = run_oracle_steps(cfg, w, "bf16", segs, np.float32)
    alt = run_oracle_steps(cfg, w, "bf16", segs, np.float64)
    proj.memory_cache = []
    worst = 0.0
    errs, floors = [], []
    for t, s_ in enumerate(segs):
        cache, _ = proj(to_dev(s_))
        assert len(cache) == min(t + 1, 10)
        got = to_np(cache[-1])
        err, floor = O.rel_l2(got, ref[t][0][-1]), O.rel_l2(alt[t][0][-1], ref[t][0][-1])
        worst = max(worst, err / chain_tol(floor))
        assert err < chain_tol(floor), (t, err, floor)
        errs.append(err)
        floors.append(floor)
        if t in (0, 15, 31):                               # no bias: the mean signed error is within 4 standard errors of 0
            d = (got - ref[t][0][-1]).astype(np.float64).reshape(-1)
            assert abs(d.mean()) < 4.0 * d.std() / math.sqrt(d.size) + 1e-7, (t, d.mean(), d.std())
    # no drift: over 32 steps the distance to the oracle grows no faster than the oracle's own summation-order noise
    # (float64- against float32-accumulating oracle) does
    assert errs[-1] / errs[0] <= 1.5 * floors[-1] / floors[0] + 0.5, (errs[0], errs[-1], floors[0], floors[-1])
    assert max(errs[16:]) <= 2.0 * max(errs[:16]) + 1e-3
    for i in range(10):                                   # the whole FIFO after three wraps, oldest first
        assert O.rel_l2(to_np(cache[i]), ref[-1][0][i]) < chain_tol(O.rel_l2(alt[-1][0][i], ref[-1][0][i]))
    print(f"1024-frame chain: worst error / tolerance = {worst:.2f}")

    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    model, _ = _tiny_host(cfg, O.make_weights(cfg, seed=62))
    T = 1024
    idx = torch.arange(T) % 600
    x = to_dev(O.bf16_round(O.hash_normal_like((T, 196, 1024), 6200)))
    mp = (torch.randn((10, 1024), device="cuda") * 0.02).bfloat16()
    fp = (torch.randn((9, 1024), device="cuda") * 0.02).bfloat16()
    eager, info = arch.video_memory_tokens(model, x, idx, mp, fp, model.image_newline)
    assert info["num_memories"] == 10 and eager.shape[0] == arch.video_token_rows(T, 8)
    assert torch.isfinite(eager.float()).all()
    g = arch.GraphedVideoMemory(model, T, idx)
    assert torch.equal(g(x, mp, fp, model.image_newline), eager)


def test_noop_module_move_keeps_the_engine():
    """The reference calls `recurrent_memory_transformer.to(self.device)` on every forward (llava_arch.py:530): a move
    that changes nothing keeps the engine (no re-packing, no re-allocation); a real one (dtype) drops it."""
    cfg = O.PathConfig(hidden=256, heads=2, mem_tokens=2, depth=2)
    proj = make_projector(cfg, O.make_weights(cfg, seed=77), "bf16")
    seg = to_dev(O.bf16_round(O.hash_normal_like((2, 196, 256), 7700)))
    proj.memory_cache = []
    a = proj(seg)[0][-1].clone()
    eng = proj._engine
    assert proj.to("cuda") is proj and proj._engine is eng and len(proj.memory_cache) == 1
    proj.memory_cache = []
    assert torch.equal(proj(seg)[0][-1], a) and proj._engine is eng
    proj.to(torch.float16)
    assert proj._engine is None and proj.memory_cache == []


def test_data_copy_parameter_update_is_seen_by_the_engine():
    """DeepSpeed ZeRO-1/2 write updated bf16 partitions back with `p.data.copy_()`, which changes neither `p._version`
    nor the pointer.  The engine holds COPIES of some parameters (fp32 biases / LayerNorm affines, mem0, the packed K/V
    matrices), so a video that starts after the module has been in training mode re-packs: the inference path then
    follows the new weights (checked against the oracle with the new weights)."""
    cfg = O.PathConfig(hidden=256, heads=2, mem_tokens=2, depth=2)
    w0 = O.make_weights(cfg, seed=31)
    w1 = O.make_weights(cfg, seed=32)
    proj = make_projector(cfg, w0, "bf16")
    segs = [O.bf16_round(O.hash_normal_like((f, 196, 256), 3100 + t)) for t, f in enumerate((2, 1))]
    proj.memory_cache = []
    for s_ in segs:
        old = proj(to_dev(s_))[0][-1].clone()
    proj.train()                                           # a training phase ...
    pfx = "recurrent_memory_transformer."
    with torch.no_grad():
        for k, p_ in proj.named_parameters():
            v0 = p_._version
            p_.data.copy_(torch.from_numpy(w1[pfx + k]).to(p_.dtype))      # ... updates every parameter through .data
            assert p_._version == v0
    proj.eval()
    proj.memory_cache = []
    for s_ in segs:
        new = proj(to_dev(s_))[0][-1]
    ref = run_oracle_steps(cfg, w1, "bf16", segs, np.float32)[-1][0][-1]
    alt = run_oracle_steps(cfg, w1, "bf16", segs, np.float64)[-1][0][-1]
    assert O.rel_l2(to_np(new), ref) < chain_tol(O.rel_l2(alt, ref))
    assert O.rel_l2(to_np(old), ref) > 0.1                 # the old weights give something else entirely
    eng = proj._engine
    proj.memory_cache = []                                 # eval mode, nothing happened since the pack: no re-pack
    ver = eng.version
    proj(to_dev(segs[0]))
    assert eng.version is ver or eng.version == ver
