"""Round 3: the row batch (several videos stepped together, `mavlm_config.batch`) and the frame masses on the scheduled
(stream-K) attention forward - HIP kernels through the C ABI against the oracle / against the single-video engine."""
import types

import numpy as np
import pytest
import torch

import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd import _capi as capi
from memory_augmented_vlm_amd import _ops as ops
from memory_augmented_vlm_amd.model.memory_module.MemoryController import BatchedProjector
from memory_augmented_vlm_amd.model import llava_arch as arch
from oracle import memory_path as O
from gpu_util import to_dev, to_np, load_oracle_weights, DT

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(autouse=True)
def _inference_path():
    with torch.no_grad():
        yield


def _host(cfg: O.PathConfig, w, mode="bf16", cap=10, vocab=48900):
    class Base(torch.nn.Module):
        def __init__(self, config):
            super().__init__()
            self.embed_tokens = torch.nn.Embedding(vocab, config.hidden_size)

    class Model(arch.LlavaMetaModel, Base):
        pass

    hf = types.SimpleNamespace(hidden_size=cfg.hidden, num_memory_tokens=cfg.mem_tokens, memory_cache_cap=cap)
    model = Model(hf).eval()
    # LlavaMetaModel hard-codes 8 heads (llava_arch.py:122); the small test shapes use fewer
    if cfg.heads != 8:
        c = model.recurrent_memory_transformer.config
        c.mm_num_attention_heads = cfg.heads
        from memory_augmented_vlm_amd.model.memory_module.MemoryController import TransformerProjector
        model.recurrent_memory_transformer = TransformerProjector(c)
        model.recurrent_memory_transformer.bind_fuser(model.memory_fuser, model.token_type_embedding)
    model.image_newline = torch.nn.Parameter(torch.zeros(cfg.hidden))
    load_oracle_weights(model, w)
    return model.to("cuda").to(DT[mode])


def _prompts(D, mode="bf16"):
    g = torch.Generator(device="cpu").manual_seed(5)
    mp = torch.randn((10, D), generator=g).to("cuda").to(DT[mode])
    fp = torch.randn((9, D), generator=g).to("cuda").to(DT[mode])
    return mp, fp


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_row_batch_bit_identical_when_the_schedules_coincide(mode):
    """Three videos stepped as one row batch == three single-video runs, bit for bit, on a shape where both take the same
    attention schedule (few key tiles: the plain grid; nothing is split).  Everything the batch adds is exercised: the
    stacked [B*R, D] GEMM / LayerNorm operands, the slot-major memory ring, the scattered outputs of the evolution K/V
    projection and of the fuser's second GEMM, per-video K/V strides in the attention, the frame masses per video, a
    FIFO that wraps (cap 2, 4 chunks), a ragged last chunk."""
    cfg = O.PathConfig(hidden=256, heads=2, mem_tokens=2, depth=2)
    w = O.make_weights(cfg, seed=61)
    model = _host(cfg, w, mode, cap=2)
    rm = model.recurrent_memory_transformer
    B, T = 3, 14
    r = O.rounder(mode)
    vids = [to_dev(r(O.hash_normal_like((T, 196, 256), 6100 + b)), mode) for b in range(B)]
    idx = torch.arange(T) * 2
    mp, fp = _prompts(256, mode)
    singles, sscores = [], []
    for b in range(B):
        n0 = len(rm.frame_attn_scores)
        singles.append(arch.video_memory_tokens(model, vids[b], idx, mp, fp, model.image_newline, chunk=4)[0].clone())
        sscores.append([s.clone() for s in rm.frame_attn_scores[n0:]])
    bp = BatchedProjector(rm, B)
    toks, info = arch.video_memory_tokens_batched(model, bp, vids, idx, mp, fp, model.image_newline, chunk=4)
    torch.cuda.synchronize()
    assert toks.shape == (B,) + tuple(singles[0].shape) and info["num_memories"] == 2
    for b in range(B):
        assert torch.equal(toks[b], singles[b]), f"video {b}"
        for c, sc in enumerate(info["frame_scores"]):
            assert torch.equal(sc[b], sscores[b][c]), f"video {b} chunk {c}"
        for got, ref in zip(bp.memory_cache(b), rm.memory_cache if b == B - 1 else []):
            assert torch.equal(got, ref)
    # the frame-dropout branch and a second batch through the same state (reset)
    toks2, _ = arch.video_memory_tokens_batched(model, bp, vids[::-1], idx, mp, fp, model.image_newline, with_frames=False,
                                                chunk=4)
    a, b_ = info["memory_rows"]
    assert toks2.shape[1] == b_ + 1
    for b in range(B):
        assert torch.equal(toks2[b], singles[B - 1 - b][:b_ + 1])


def test_next_chunk_projection_on_a_side_stream_changes_nothing():
    """Round 4: at few memory tokens `video_memory_tokens` enqueues the NEXT chunk's K/V projection on a side stream before each step
    (`mavlm_project_chunk_ahead`: two chunk K/V buffers, event-ordered).  Same bits as the plain loop (`PROJECT_AHEAD = False`), every
    step but the first finds its projection (mavlm_prefetch_hits), a ragged last chunk and back-to-back videos included; a chunk that
    is NOT the announced one discards the projection and still gives the right result."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=62)
    model = _host(cfg, w, "bf16")
    rm = model.recurrent_memory_transformer
    T = 100                                                    # 32 + 32 + 32 + 4 frames
    vids = [to_dev(O.bf16_round(O.hash_normal_like((T, 196, 1024), 6200 + b)), "bf16") for b in range(2)]
    idx = torch.arange(T)
    mp, fp = _prompts(1024)
    lib = capi.lib()
    assert rm.ahead_ok(force=True) and not rm.ahead_ok()       # (automatic only at hidden >= 2048: no gain at D = 1024)
    prev = arch.PROJECT_AHEAD
    try:
        arch.PROJECT_AHEAD = False
        plain = [arch.video_memory_tokens(model, v, idx, mp, fp, model.image_newline)[0].clone() for v in vids]
        pscores = [s.clone() for s in rm.frame_attn_scores[-4:]]
        arch.PROJECT_AHEAD = True
        eng = rm.engine(vids[0].device, vids[0].dtype)
        h0 = lib.mavlm_prefetch_hits(eng.ctx)
        ahead = [arch.video_memory_tokens(model, v, idx, mp, fp, model.image_newline)[0].clone() for v in vids]
        torch.cuda.synchronize()
        assert lib.mavlm_prefetch_hits(eng.ctx) - h0 == 2 * 3
        for a, b in zip(ahead, plain):
            assert torch.equal(a, b)
        for a, b in zip(rm.frame_attn_scores[-4:], pscores):
            assert torch.equal(a, b)
        # an announced chunk that does not come: the step projects its own chunk
        x = vids[0]
        rm.memory_cache = []
        rm(x[:32])
        ref1 = rm(x[32:64])[0][-1].clone()
        rm.memory_cache = []
        rm.project_ahead(x[64:96])                              # announces the wrong chunk for step 1
        rm(x[:32])
        got1 = rm(x[32:64])[0][-1].clone()
        torch.cuda.synchronize()
        assert torch.equal(got1, ref1)
    finally:
        arch.PROJECT_AHEAD = prev


def test_next_chunk_projection_is_automatic_at_the_ov7b_width():
    """... and at hidden 3584 with 8 memory tokens it is what `video_memory_tokens` does by itself (measured +1.3 %): same tokens as
    with `PROJECT_AHEAD = False`, two of three steps find their projection."""
    cfg = O.PathConfig(hidden=3584, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=63)
    model = _host(cfg, w, "bf16")
    rm = model.recurrent_memory_transformer
    assert rm.ahead_ok()
    T = 11
    vid = to_dev(O.bf16_round(O.hash_normal_like((T, 196, 3584), 6300)), "bf16")
    idx = torch.arange(T)
    mp, fp = _prompts(3584)
    lib = capi.lib()
    prev = arch.PROJECT_AHEAD
    try:
        arch.PROJECT_AHEAD = False
        plain = arch.video_memory_tokens(model, vid, idx, mp, fp, model.image_newline, chunk=4)[0].clone()
        arch.PROJECT_AHEAD = None
        eng = rm.engine(vid.device, vid.dtype)
        h0 = lib.mavlm_prefetch_hits(eng.ctx)
        auto = arch.video_memory_tokens(model, vid, idx, mp, fp, model.image_newline, chunk=4)[0].clone()
        torch.cuda.synchronize()
        assert lib.mavlm_prefetch_hits(eng.ctx) - h0 == 2
        assert torch.equal(auto, plain)
    finally:
        arch.PROJECT_AHEAD = prev


def test_row_batch_checkpoint_shape_vs_oracle():
    """The reference-default shape (8 memory tokens, D = 1024, 8 heads), five videos in one row batch, 2 chunks of 32 frames:
    40 (video, head) pairs x 7 query blocks = 280 eight-wave units on 256 workgroups - the levelled stream-K schedule with
    the frame masses riding on it, cut units in the last video.  Video 0 (whole units) and video 4 (cut units, entries per
    piece merged in the combine kernel) against the oracle with the batch's plan mirrored; every video against the
    single-video engine within the 16-bit rounding of two schedules."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=62)
    model = _host(cfg, w)
    rm = model.recurrent_memory_transformer
    B, T = 5, 64
    info_ = (capi.i32 * 4)()
    capi.check(capi.lib().mavlm_attention_plan(1568, 6272, 8 * B, info_), "plan")
    assert info_[1] == 256 and info_[0] == 8 and info_[2] == 2          # 8-wave stream-K, two levels (16 + 8 cut units)
    xs = [O.bf16_round(O.hash_normal_like((T, 196, 1024), 6200 + b)) for b in range(B)]
    vids = [to_dev(x) for x in xs]
    idx = torch.arange(T)
    mp, fp = _prompts(1024)
    bp = BatchedProjector(rm, B)
    toks, info = arch.video_memory_tokens_batched(model, bp, vids, idx, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    a, b_ = info["memory_rows"]
    emb = np.zeros((48900, 1024), np.float32)
    emb[list(O.MEM_PROMPT_IDS)] = to_np(mp)
    emb[list(O.FRAME_PROMPT_IDS)] = to_np(fp)
    for b in (0, B - 1):
        O.ROW_BATCH = (b, B)
        try:
            ref, parts = O.video_tokens(xs[b], idx.numpy(), cfg, w, emb, "bf16", return_parts=True)
            with O.accumulate_in(np.float64):
                ref64 = O.video_tokens(xs[b], idx.numpy(), cfg, w, emb, "bf16")
        finally:
            O.ROW_BATCH = (0, 1)
        floor = O.rel_l2(ref64[a:b_], ref[a:b_])
        err = O.rel_l2(to_np(toks[b])[a:b_], ref[a:b_])
        print(f"row batch, video {b}: fused memory tokens HIP vs oracle {err:.2e} (floor {floor:.2e})")
        assert err < max(TOL, 2.0 * floor)
        for c, sc in enumerate(info["frame_scores"]):
            assert O.rel_l2(to_np(sc[b]), parts["frame_scores"][c]) < 5e-3
    for b in range(B):
        single = arch.video_memory_tokens(model, vids[b], idx, mp, fp, model.image_newline)[0]
        assert O.rel_l2(to_np(toks[b]), to_np(single)) < 6e-3
        assert torch.equal(toks[b][:a], single[:a]) and torch.equal(toks[b][b_:], single[b_:])     # copies / gathers


def test_row_batch_wide_heads_ov7b_width():
    """The OneVision-7B width (D = 3584, head_dim 448) in a row batch: the weight-shared GEMMs / LayerNorms run over the
    stacked rows, the wide-head attention once per video - every video's tokens equal the single-video engine's bit for bit
    where the GEMM kernels coincide, else within the rounding of the split-K form the single video takes."""
    cfg = O.PathConfig(hidden=3584, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=65)
    model = _host(cfg, w)
    rm = model.recurrent_memory_transformer
    B, T = 3, 40
    vids = [to_dev(O.bf16_round(O.hash_normal_like((T, 196, 3584), 6500 + b))) for b in range(B)]
    idx = torch.arange(T)
    mp, fp = _prompts(3584)
    bp = BatchedProjector(rm, B)
    toks, info = arch.video_memory_tokens_batched(model, bp, vids, idx, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    a, b_ = info["memory_rows"]
    for b in range(B):
        n0 = len(rm.frame_attn_scores)
        single = arch.video_memory_tokens(model, vids[b], idx, mp, fp, model.image_newline)[0]
        err = O.rel_l2(to_np(toks[b])[a:b_], to_np(single)[a:b_])
        print(f"OV-7B width, row batch video {b}: fused rows vs single-video engine {err:.2e}")
        assert err < 6e-3
        assert torch.equal(toks[b][:a], single[:a]) and torch.equal(toks[b][b_:], single[b_:])
        for c_, sc in enumerate(info["frame_scores"]):
            assert O.rel_l2(to_np(sc[b]), to_np(rm.frame_attn_scores[n0 + c_])) < 5e-3


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("mode,R,F,P,H", [("bf16", 8320, 10, 64, 8), ("fp16", 8330, 3, 196, 8), ("bf16", 1100, 6, 100, 64)])
def test_attention_frames_on_the_stream_k_schedule(mode, R, F, P, H, waves, monkeypatch, request):
    """The frame-score variant of the forward on the levelled stream-K schedule: the pieces of a cut unit leave (a, m)
    entries for the parts of the frames they saw, the combine kernel - which knows the merged log-sum-exp - turns them into
    partial frame sums.  Context and log-sum-exp are BIT-identical to the plain forward of the same schedule (asking for
    the scores has no side effect), the scores match the oracle's column sums; frame boundaries inside pieces, pieces
    inside frames, fewer key tiles than pieces (empty pieces), ragged query blocks; 4- and 8-wave workgroups."""
    lib = capi.lib()
    capi.check(lib.mavlm_set_attention_streamk_min_tiles(2), "min tiles")
    capi.check(lib.mavlm_set_attention_streamk_waves(waves), "waves")
    request.addfinalizer(lambda: (lib.mavlm_set_attention_streamk_min_tiles(64), lib.mavlm_set_attention_streamk_waves(0)))
    monkeypatch.setattr(O, "STREAMK_MIN_TILES", 2)
    monkeypatch.setattr(O, "STREAMK_WAVES", waves)
    S = F * P
    info = (capi.i32 * 4)()
    capi.check(lib.mavlm_attention_plan(R, S, H, info), "plan")
    assert info[1] > 0 and info[0] == waves, "the shape must take the stream-K schedule"
    r = O.rounder(mode)
    q = r(O.hash_normal_like((R, H * 128), 71)) * 2.0
    k = r(O.hash_normal_like((S, H * 128), 72))
    v = r(O.hash_normal_like((S, H * 128), 73))
    q[R - 7] *= 8.0                                    # a row of the last (cut) units whose maximum jumps: rescale of the mass
    dq, dk, dv = to_dev(q, mode), to_dev(k, mode), to_dev(v, mode)
    got, lse, scores = ops.attention_frames(dq, dk, dv, H, P, want_lse=True)
    plain, lse_p = ops.attention(dq, dk, dv, H, want_lse=True)
    assert torch.equal(got, plain) and torch.equal(lse, lse_p)
    ctx, lse2, col, _ = O.attention_heads(q, k, v, H, mode, want_colsum=True)
    assert O.rel_l2(to_np(got), r(ctx)) < TOL
    ref = col.astype(np.float64).sum(0).reshape(F, P).mean(1)
    assert O.rel_l2(to_np(scores), ref) < TOL and abs(float(scores.sum()) * P - H * R) < 1e-3 * H * R
    for _ in range(2):
        assert torch.equal(ops.attention_frames(dq, dk, dv, H, P)[2], scores)


@pytest.mark.parametrize("R", [1568, 300, 4100 + 64])
def test_frame_entries_have_one_writer(R):
    """Query rows past R are clamped duplicates of row R-1.  When R % 128 leaves whole waves past R, those waves must not
    store frame entries: a duplicate wave can take a rescale the owning wave does not (a row of wave 0 - not row R-1 -
    whose maximum jumps by more than 2^8 in a late tile) and would race with it.  Their stores go to an offset behind the
    buffer descriptor's end.  Scores equal the oracle's and repeat bit for bit."""
    H, P, F = 8, 196, 6
    S = F * P
    assert 0 < R % 128 <= 96
    r = O.rounder("bf16")
    q = r(O.hash_normal_like((R, H * 128), 91))
    k = r(O.hash_normal_like((S, H * 128), 92))
    v = r(O.hash_normal_like((S, H * 128), 93))
    row = (R // 128) * 128 + 5                        # wave 0 of the last block, not the last row
    assert row < R - 1
    # its score against the keys of a late tile exceeds everything before by far more than 2^8 (log2 domain)
    k[S - 100:S - 90] = r(q[row][None, :] * 6.0)
    dq, dk, dv = to_dev(q), to_dev(k), to_dev(v)
    got, lse, scores = ops.attention_frames(dq, dk, dv, H, P, want_lse=True)
    _, _, col, _ = O.attention_heads(q, k, v, H, "bf16", want_colsum=True, plain=True)
    ref = col.astype(np.float64).sum(0).reshape(F, P).mean(1)
    assert O.rel_l2(to_np(scores), ref) < TOL
    for _ in range(20):
        assert torch.equal(ops.attention_frames(dq, dk, dv, H, P)[2], scores)


def test_pool_row_batches():
    """MemoryPathPool(batch=B): consecutive videos of one length run as row batches on the pool's streams, the rest through
    the single-video slots; every video's block equals the single-video result within the rounding of two attention
    schedules, and identical videos in different batch positions give identical tokens."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=63)
    model = _host(cfg, w)
    mp, fp = _prompts(1024)
    x40 = [to_dev(O.bf16_round(O.hash_normal_like((40, 196, 1024), 6300 + i))) for i in range(4)]
    x33 = to_dev(O.bf16_round(O.hash_normal_like((33, 196, 1024), 6310)))
    vids = [(x40[0], torch.arange(40)), (x40[1], torch.arange(40)), (x33, torch.arange(33)),
            (x40[2], torch.arange(40)), (x40[0], torch.arange(40)), (x40[3], torch.arange(40))]
    serial = [arch.video_memory_tokens(model, v, i, mp, fp, model.image_newline)[0].clone() for v, i in vids]
    pool = arch.MemoryPathPool(model, 2, batch=2)
    outs = pool.run(vids, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    for a, b in zip(serial, outs):
        assert a.shape == b.shape and O.rel_l2(to_np(b), to_np(a)) < 6e-3
    assert torch.equal(outs[2], serial[2])                  # the odd one out ran through a single-video slot


def test_row_batch_refusals():
    """Loud errors: batch sizes out of range, a row shard combined with a row batch; the single-video entry points refuse a
    batched context."""
    c = capi.Config(hidden=1024, heads=8, patches=196, mem_tokens=8, depth=2, inter=4096, cache_cap=10,
                    max_chunk_frames=32, dtype=0, eps=1e-12, batch=65)
    h = capi.vp()
    assert capi.lib().mavlm_create(c, h) == capi.E_ARG
    c.batch, c.q_tokens = 2, 4
    assert capi.lib().mavlm_create(c, h) == capi.E_ARG
    c.q_tokens = 0
    c.batch = 2
    assert capi.lib().mavlm_create(c, h) == 0 and capi.lib().mavlm_batch(h) == 2
    x = torch.zeros(8, device="cuda")
    assert capi.lib().mavlm_step(h, x.data_ptr(), 1, 0, 0, 0) == capi.E_STATE
    capi.lib().mavlm_destroy(h)


# ---- fused dense + residual + LayerNorm epilogue (mavlm_linear_ln) ----------------------------------------------------
def _ln_case(M, N, K, mode, seed):
    r = O.rounder(mode)
    x = r(O.hash_normal_like((M, K), seed))
    w = r(O.hash_uniform((N, K), seed + 1, -1 / np.sqrt(K), 1 / np.sqrt(K)))
    b = O.hash_uniform((N,), seed + 2, -0.1, 0.1).astype(np.float32)
    res = r(O.hash_normal_like((M, N), seed + 3))
    g = (1.0 + O.hash_uniform((N,), seed + 4, -0.1, 0.1)).astype(np.float32)
    be = O.hash_uniform((N,), seed + 5, -0.1, 0.1).astype(np.float32)
    return x, w, b, res, g, be


@pytest.mark.parametrize("mode,M,N,K", [("bf16", 12544, 1024, 1024), ("bf16", 25088 + 37, 1024, 4096), ("fp16", 12544, 1024, 1024),
                                        ("bf16", 6272, 3584, 512), ("bf16", 13000, 1024, 64), ("bf16", 50000, 256, 128 * 3)])
def test_fused_dense_residual_layernorm(mode, M, N, K, request):
    """mavlm_linear_ln: the Residual block (MemoryController.py:20-29) as ONE kernel - the N / 256 workgroups of a row block
    exchange their per-row (mean, centred sum of squares) through {epoch, value} granules - against the oracle's LayerNorm of
    the oracle's dense output and against the two-kernel form (same fp32 inputs, statistics added in another order); the
    optional fp32 dense output equals the plain GEMM's bit for bit; N = 1024 (4 tiles per row), 3584 (14: the OV-7B width)
    and 256 (1: no partner), ragged M, a single K-tile; bit-reproducible."""
    from gpu_util import f32_dev
    lib = capi.lib()
    on = 2 if N > 1024 else 1            # rows wider than 1024 columns: admitted by the test hook only (slower than two kernels)
    request.addfinalizer(lambda: lib.mavlm_set_fused_layernorm(1))
    if N > 1024:
        assert lib.mavlm_linear_ln_ws_bytes(M, N, K) == 0
    capi.check(lib.mavlm_set_fused_layernorm(on), "hook")
    assert lib.mavlm_linear_ln_ws_bytes(M, N, K) > 0 and lib.mavlm_linear_ln_ws_bytes(1568, N, K) == 0
    x, w, b, res, g, be = _ln_case(M, N, K, mode, 300 + N)
    dx, dw, dres = to_dev(x, mode), to_dev(w, mode), to_dev(res, mode)
    db, dg, dbe = f32_dev(b), f32_dev(g), f32_dev(be)
    out, pre = ops.linear_residual_layernorm(dx, dw, db, dres, dg, dbe, 1e-12, want_pre=True)
    plain_pre = ops.linear(dx, dw, db, capi.EPI_F32)
    assert torch.equal(pre, plain_pre)
    capi.check(lib.mavlm_set_fused_layernorm(0), "hook")
    two, _ = ops.linear_residual_layernorm(dx, dw, db, dres, dg, dbe, 1e-12)
    capi.check(lib.mavlm_set_fused_layernorm(on), "hook")
    r = O.rounder(mode)
    ref = r(O.layernorm(O.linear(x, w, b) + res, g, be, 1e-12))
    err, err2, d12 = O.rel_l2(to_np(out), ref), O.rel_l2(to_np(two), ref), O.rel_l2(to_np(out), to_np(two))
    print(f"fused LN {mode} {M}x{N}x{K}: vs oracle {err:.2e} (two-kernel form {err2:.2e}), fused vs two-kernel {d12:.2e}")
    assert err < TOL and err <= err2 * 1.2 + 1e-6 and d12 < 5e-4
    for _ in range(5):
        assert torch.equal(ops.linear_residual_layernorm(dx, dw, db, dres, dg, dbe, 1e-12)[0], out)


def test_fused_layernorm_back_to_back_launches_and_graph_replay():
    """The exchange granules carry an epoch from a launch counter in memory (nothing is re-zeroed between launches): many
    launches back to back on alternating inputs each give their own result (a stale granule of the previous launch would
    show as the other input's statistics), and a captured launch replays correctly - the counter lives in memory, not in
    the kernel arguments."""
    from gpu_util import f32_dev
    M, N, K = 12544, 1024, 1024
    cases = []
    for sd in (700, 800):
        x, w, b, res, g, be = _ln_case(M, N, K, "bf16", sd)
        res = res * (3.0 if sd == 800 else 1.0) + (5.0 if sd == 800 else 0.0)       # very different row statistics
        args = (to_dev(x), to_dev(w), f32_dev(b), to_dev(res), f32_dev(g), f32_dev(be), 1e-12)
        cases.append((args, ops.linear_residual_layernorm(*args)[0].clone()))
    assert not torch.equal(cases[0][1], cases[1][1])
    outs = []
    for i in range(40):
        outs.append(ops.linear_residual_layernorm(*cases[i & 1][0])[0])
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, cases[i & 1][1]), i
    # hipGraph: capture one launch, replay it on changing inputs (static buffers)
    a0 = cases[0][0]
    sx, sres = a0[0].clone(), a0[3].clone()
    sout = torch.empty_like(cases[0][1])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.linear_residual_layernorm(sx, a0[1], a0[2], sres, a0[4], a0[5], 1e-12, out=sout)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    # (torch.cuda.graph captures on a stream of its own by default, where the operator has no scratch yet - it would raise
    # "first use inside a graph capture"; capture on the warmed-up stream: the scratch is per (device, stream))
    with torch.cuda.graph(gr, stream=side):
        ops.linear_residual_layernorm(sx, a0[1], a0[2], sres, a0[4], a0[5], 1e-12, out=sout)
    for rep in range(6):
        src = cases[rep & 1]
        sx.copy_(src[0][0])
        sres.copy_(src[0][3])
        gr.replay()
        torch.cuda.synchronize()
        if rep & 1:      # (input x / res of case 1 through the weights of case 0: compare with an eager launch)
            want = ops.linear_residual_layernorm(sx, a0[1], a0[2], sres, a0[4], a0[5], 1e-12)[0]
        else:
            want = cases[0][1]
        assert torch.equal(sout, want), rep


def test_fused_layernorm_in_the_step_training_equals_inference():
    """64 memory tokens (R = 12 544: the Residual blocks of the step run as the fused kernel): the training path's forward
    (autograd Functions over the same operators) is bit-identical to the inference step, the fused step matches the
    two-kernel step within the rounding of a different summation order, and gradients flow."""
    from test_gpu_path import make_projector
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=64, depth=2)
    w = O.make_weights(cfg, seed=64)
    proj = make_projector(cfg, w)
    segs = [to_dev(O.bf16_round(O.hash_normal_like((f, 196, 1024), 6400 + t))) for t, f in enumerate((3, 2))]
    proj.memory_cache = []
    for s_ in segs:
        cache, _ = proj(s_)
    infer = [c.clone() for c in cache]
    launches, timeouts = proj._engine.ln_exchange_status()
    assert launches == 2 * (2 * 2) + 1 and timeouts == 0          # 2 Residual blocks per layer, 2 layers, 2 steps + 1 evolution
    lib = capi.lib()
    try:
        capi.check(lib.mavlm_set_fused_layernorm(0), "hook")
        proj._engine = None                   # a context snapshots the hook at mavlm_create: a new engine takes the new form
        proj.memory_cache = []
        for s_ in segs:
            cache, _ = proj(s_)
        two = [c.clone() for c in cache]
    finally:
        lib.mavlm_set_fused_layernorm(1)
    assert proj._engine.ln_exchange_status() is None            # that engine never fuses, whatever the hook says now
    proj._engine = None
    for a, b in zip(infer, two):
        assert O.rel_l2(to_np(a), to_np(b)) < 3e-3
    with torch.enable_grad():
        for p_ in proj.parameters():
            p_.requires_grad_(True)
        proj.memory_cache = []
        for s_ in segs:
            cache, _ = proj(s_)
        for a, b in zip(cache, infer):
            assert torch.equal(a.detach(), b)
        sum((c.float() ** 2).mean() for c in cache).backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad.float()).all() for p_ in proj.parameters())
    proj.memory_cache = []


def test_bench_launch_configuration_vs_reference_golden():
    """bench.py's own launch configuration against the REFERENCE: the G7 `m64f32` video (= the bench workload:
    64 memory tokens, two 32-frame chunks, D = 1024) through `BatchedProjector(rm, 2)` inside `MemoryPathPool(model, 2, batch=2)`
    with BOTH streams busy - 16 (video, head) pairs per attention launch on the 1-level stream-K plan with the frame masses on
    the cut units, the fused Residual kernel over 25 088 rows, row-scattered GEMM outputs, two launch sequences interleaving
    on the chip.  Every one of the 4 videos passes the G7 gates (at least as close to the reference's fp32 run as the
    reference's own bf16 run, unbiased, no faster drift, frame scores), no exchange timeout, and the 2-stream run equals the
    1-stream run bit for bit (streams only interleave independent launch sequences)."""
    import math
    from test_gpu_path import load_golden, _g7_gates
    z, m = load_golden("g7_fullsize.npz")
    tag, M, F, steps = "m64f32", 64, 32, 2
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    model = _host(cfg, w)
    model.positional_encoding.frame_embed.zero_()              # PE add = identity: the chunks reach the projector as G7's
    mp, fp = _prompts(1024)
    x = to_dev(np.concatenate([O.bf16_round(O.hash_normal_like((F, 196, 1024), m["segseed0"] + t)) for t in range(steps)]))
    vids = [(x.clone(), torch.arange(F * steps)) for _ in range(4)]
    pool2 = arch.MemoryPathPool(model, 2, batch=2)
    outs2 = pool2.run(vids, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    assert len(pool2.bslots) == 2 and all(bs._n == steps for bs in pool2.bslots)        # both streams stepped a batch
    mems2 = []
    for k, bs in enumerate(pool2.bslots):
        launches, timeouts = bs._engine.ln_exchange_status()
        assert timeouts == 0 and launches == 9, (launches, timeouts)    # 2 x 2 x 2 formation + 1 evolution Residual blocks
        for b in range(2):
            cache = bs.memory_cache(b)
            assert len(cache) == steps
            errs, ref_errs = [], []
            for t in range(steps):
                mem = to_np(cache[t]).reshape(-1)
                e, r = _g7_gates(mem[::m["stride"]], mem, to_np(bs.frame_scores[t][b]), z, tag + "_", t, m,
                                 f"bench config, stream {k} video {b}")
                errs.append(e)
                ref_errs.append(r)
            assert errs[1] / errs[0] <= 1.25 * ref_errs[1] / ref_errs[0] + 0.05
            mems2.append([c.clone() for c in cache])
    pool1 = arch.MemoryPathPool(model, 1, batch=2)
    outs1 = pool1.run(vids, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    for a, b in zip(outs1, outs2):
        assert torch.equal(a, b)
    for b in range(2):                                          # (pool1's one slot ran both groups: it holds the last one)
        for t in range(steps):
            assert torch.equal(pool1.bslots[0].memory_cache(b)[t], mems2[2 + b][t])
    assert torch.equal(outs2[0], outs2[2]) and torch.equal(outs2[1], outs2[3])          # same video, other stream


def test_fused_layernorm_operator_on_two_streams():
    """the operator-level fused dense + residual + LayerNorm keeps ONE exchange scratch per (device, stream) - two
    streams launching it concurrently (same epoch, same granule slots on a shared scratch) would cross-contaminate the row
    statistics.  Two streams, different inputs, interleaved launches: each stream's results equal its own serial result bit
    for bit and agree with the two-kernel form; a larger shape later gets a new scratch, the outgrown one stays alive."""
    from gpu_util import f32_dev
    lib = capi.lib()
    cases = []
    for i in range(2):
        x, w, b, res, g, be = _ln_case(12544, 1024, 1024, "bf16", 900 + 10 * i)
        cases.append((to_dev(x), to_dev(w), f32_dev(b), to_dev(res), f32_dev(g), f32_dev(be)))
    serial = [ops.linear_residual_layernorm(*c, 1e-12)[0].clone() for c in cases]
    try:
        lib.mavlm_set_fused_layernorm(0)
        two = [ops.linear_residual_layernorm(*c, 1e-12)[0].clone() for c in cases]
    finally:
        lib.mavlm_set_fused_layernorm(1)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for rep in range(12):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i].append(ops.linear_residual_layernorm(*cases[i], 1e-12)[0])
    torch.cuda.synchronize()
    keys = [k for k in ops._LN_WS if k[0] == torch.cuda.current_device()]
    assert len({k[1] for k in keys}) >= 3                       # default stream + the two side streams: a scratch each
    for i in range(2):
        for o in outs[i]:
            assert torch.equal(o, serial[i])
        assert O.rel_l2(to_np(serial[i]), to_np(two[i])) < 5e-4
    # a shape that outgrows the scratch: a new one, the old tensor is kept (captured graphs may hold its pointer)
    kept = len(ops._LN_WS_KEEP)
    x, w, b, res, g, be = _ln_case(150016, 1024, 64, "bf16", 990)
    big = ops.linear_residual_layernorm(to_dev(x), to_dev(w), f32_dev(b), to_dev(res), f32_dev(g), f32_dev(be), 1e-12)[0]
    ref = O.rounder("bf16")(O.layernorm(O.linear(x, w, b) + res, g, be, 1e-12))
    assert O.rel_l2(to_np(big), ref) < TOL and len(ops._LN_WS_KEEP) == kept + 1
    assert torch.equal(ops.linear_residual_layernorm(*cases[0], 1e-12)[0], serial[0])


def test_fused_layernorm_timeout_is_raised_by_the_product_path():
    """the product path notices a timed-out exchange.  The timeout word of the engine's workspace is set by
    hand; the next `memory_cache = []` posts the asynchronous probe (mavlm_ln_status_async: no synchronisation) and the one
    after it raises MavlmError; the flag is cleared by the probe, so the video after that runs again."""
    from test_gpu_path import make_projector
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=64, depth=2)
    proj = make_projector(cfg, O.make_weights(cfg, seed=65))
    seg = to_dev(O.bf16_round(O.hash_normal_like((2, 196, 1024), 6500)))
    proj.memory_cache = []
    proj(seg)
    eng = proj._engine
    assert eng.ln_exchange_status() == (4, 0)
    off = eng.workspace_base_offset + int(capi.lib().mavlm_ln_ctl_offset(eng.ctx))
    eng.workspace[off + 8:off + 12].view(torch.int32).fill_(1)          # a workgroup gave up (never seen for real)
    proj.memory_cache = []                                              # posts the probe
    proj(seg)
    torch.cuda.synchronize()
    with pytest.raises(capi.MavlmError, match="timed out"):
        proj.memory_cache = []                                          # looks at it
    proj.memory_cache = []                                              # cleared by the probe: business as usual
    proj(seg)
    torch.cuda.synchronize()
    proj.memory_cache = []
    assert eng.ln_exchange_status()[1] == 1                             # (the count of what was seen stays visible)


def test_pool_with_more_streams_than_the_fused_layernorm_admits():
    """include/mavlm.h MAVLM_LN_MAX_STREAMS: a pool of 16 streams at 64 memory tokens must not run 16 fused Residual launches
    side by side (3 waiting workgroups per launch and XCD).  Its slots are created with mavlm_config.fused_ln = NEVER: the
    engines carve no exchange scratch and run GEMM + row LayerNorm; results within the rounding of the other summation order
    of the serial (fused) path, identical videos identical across slots."""
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=64, depth=2)
    w = O.make_weights(cfg, seed=66)
    model = _host(cfg, w)
    mp, fp = _prompts(1024)
    x = to_dev(O.bf16_round(O.hash_normal_like((34, 196, 1024), 6600)))
    vids = [(x.clone(), torch.arange(34)) for _ in range(3)]
    serial = arch.video_memory_tokens(model, x, torch.arange(34), mp, fp, model.image_newline)[0].clone()
    assert model.recurrent_memory_transformer._engine.ln_exchange_status() is not None       # the serial engine fuses
    pool = arch.MemoryPathPool(model, 16)
    assert pool.fused_ln_never and len(pool.slots) == 16
    outs = pool.run(vids, mp, fp, model.image_newline)
    torch.cuda.synchronize()
    for k in range(3):
        eng = pool.slots[k].recurrent_memory_transformer._engine
        assert eng.c.fused_ln == capi.LN_NEVER and eng.ln_exchange_status() is None
        assert torch.equal(outs[k], outs[0]) and O.rel_l2(to_np(outs[k]), to_np(serial)) < 6e-3
    assert not arch.MemoryPathPool(model, capi.LN_MAX_STREAMS).fused_ln_never


def test_project_chunk_prefetch_is_dropped_when_the_weights_change():
    """mavlm_project_chunk leaves the chunk's K/V in the workspace for the next mavlm_step; mavlm_bind_weights (a
    re-pack after a parameter update) must discard it - the step then projects again with the new weights."""
    from test_gpu_path import make_projector
    cfg = O.PathConfig(hidden=256, heads=2, mem_tokens=4, depth=2)
    proj = make_projector(cfg, O.make_weights(cfg, seed=67))
    seg = to_dev(O.bf16_round(O.hash_normal_like((3, 196, 256), 6700)))
    lib = capi.lib()
    proj.memory_cache = []
    eng = proj.engine(seg.device, seg.dtype, 3)
    capi.check(lib.mavlm_project_chunk(eng.ctx, seg.data_ptr(), 3, ops.stream_ptr()), "project")
    with torch.no_grad():
        proj.layers[0].memory_segment_fusion_attention.k_proj.weight.mul_(1.5)         # bumps the parameter version
    got = proj(seg)[0][-1].clone()                                     # engine() re-packs -> mavlm_bind_weights -> step
    proj.memory_cache = []
    fresh = proj(seg)[0][-1].clone()
    assert torch.equal(got, fresh)
    # ... and a prefetch on ANOTHER stream is not reused either (the cache is tied to the stream that produced it)
    proj.memory_cache = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        capi.check(lib.mavlm_project_chunk(eng.ctx, seg.data_ptr(), 3, ops.stream_ptr()), "project")
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(proj(seg)[0][-1], fresh)
