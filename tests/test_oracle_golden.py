"""Pins the CPU oracle (oracle/memory_path.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference on CPU).
Tolerance: fp32 restatement vs reference fp32 <= 1e-5 rel-L2; index math exact."""
import numpy as np
import pytest

from oracle import memory_path as O
from conftest import load_golden

TOL = 1e-5


def test_g1_attention_and_layer():
    z, m = load_golden("g1_attention.npz")
    cfg = O.PathConfig(hidden=m["hidden"], heads=m["heads"], mem_tokens=m["mem_tokens"], depth=m["depth"])
    w = O.make_weights(cfg, seed=m["wseed"])
    xq = O.bf16_round(O.hash_normal_like((m["R"], cfg.hidden), m["qseed"]))
    xkv = O.bf16_round(O.hash_normal_like((m["S"], cfg.hidden), m["kvseed"]))
    pfx = "recurrent_memory_transformer.layers.0"
    out, colsum, probs = O.mha(xq, xkv, w, pfx + ".memory_segment_fusion_attention", cfg, "fp32",
                               want_colsum=True, want_probs=True)
    assert O.rel_l2(out, z["out"]) < TOL
    assert O.rel_l2(colsum, z["colsum"]) < TOL
    assert O.rel_l2(probs[0, :4], z["probs_h0_rows"]) < TOL
    lay, _ = O.transformer_layer(xq, xkv, w, pfx, cfg, "fp32", False)
    assert O.rel_l2(lay, z["layer_out"]) < TOL
    # sum of all probabilities = H * R  (MemoryController.py:135: softmax rows sum to one)
    assert abs(colsum.sum() - cfg.heads * m["R"]) < 1e-2


def test_g3_recurrent_steps():
    z, m = load_golden("g3_recurrent.npz")
    cfg = O.PathConfig(hidden=m["hidden"], heads=m["heads"], mem_tokens=m["mem_tokens"], depth=m["depth"])
    w = O.make_weights(cfg, seed=m["wseed"])
    rm = O.RecurrentMemory(cfg, w, "fp32")
    rm.reset()
    for t in range(m["steps"]):
        seg = O.bf16_round(O.hash_normal_like((m["F"], 196, cfg.hidden), m["segseed0"] + t))
        cache, scores = rm.step(seg)
        assert len(cache) == t + 1
        mem = cache[-1]
        ref = z[f"mem{t}"]
        got = mem if t == 3 else mem[:, ::m["rowstride_first3"], :]
        assert O.rel_l2(got, ref) < TOL, t
        assert abs(mem.astype(np.float64).sum() - float(z[f"mem{t}_sum"])) < 1e-2
        assert O.rel_l2(scores[-1], z[f"score{t}"]) < TOL


def test_g3_fifo_eviction():
    z, m = load_golden("g3_fifo.npz")
    cfg = O.PathConfig(hidden=m["hidden"], heads=m["heads"], mem_tokens=m["mem_tokens"], depth=m["depth"])
    w = O.make_weights(cfg, seed=m["wseed"])
    rm = O.RecurrentMemory(cfg, w, "fp32")
    rm.reset()
    for t, f in enumerate(m["frames"]):
        seg = O.bf16_round(O.hash_normal_like((f, 196, cfg.hidden), m["segseed0"] + t))
        cache, scores = rm.step(seg)
    assert len(cache) == 10 and len(scores) == int(z["n_scores"]) == len(m["frames"])
    got = np.stack(cache)
    assert O.rel_l2(got[:, :, ::m["rowstride"], :], z["cache"]) < 2 * TOL
    np.testing.assert_allclose(got.astype(np.float64).sum(axis=(1, 2, 3)), z["cache_sums"], atol=2e-2)
    assert O.rel_l2(scores[-1], z["scores_last"]) < TOL


def test_g4_positional_encoding():
    z, m = load_golden("g4_pe.npz")
    D = m["D"]
    table = O.pe_table(600, D)
    # float32 exp/sin differ by an ulp between numpy's libm and ATen's SLEEF; the angle is
    # position(<=599) * div_term, so one ulp of div_term moves sin/cos by <= 599 * 2^-24 ~ 3.6e-5 (plus the half-ulp of the ~600 rad product, 3e-5).
    np.testing.assert_allclose(table, z["table"], rtol=0, atol=1e-4)
    t1024 = O.pe_table(600, 1024)
    np.testing.assert_allclose(t1024[[0, 1, 2, 299, 599]], z["table1024_rows"], rtol=0, atol=1e-4)
    x = O.hash_normal_like((7, 196, D), m["xseed"])
    y = O.pe_add(x, z["idx"], z["table"], "fp32")
    np.testing.assert_allclose(y, z["y"], rtol=0, atol=1e-6)
    ydef = O.pe_add(x, np.arange(7), z["table"], "fp32")
    np.testing.assert_allclose(ydef, z["ydef"], rtol=0, atol=1e-6)
    with pytest.raises(ValueError, match="exceed max_frames"):
        O.pe_add(x[:2], np.array([0, 600]), table)
    with pytest.raises(ValueError, match="negative"):
        O.pe_add(x[:2], np.array([-1, 5]), table)
    with pytest.raises(ValueError, match="Expected 3D"):
        O.pe_add(x[0], np.array([0]), table)
    assert "exceed max_frames" in m["errors"]["too_big"] and "negative" in m["errors"]["negative"]


def test_g5_index_math_exact():
    z, _ = load_golden("g5_index.npz")
    cases = sorted(int(k.split("_")[1]) for k in z.files if k.startswith("idx_"))
    assert 599 in cases and 1 in cases
    for F0 in cases:
        idx = O.subsample_indices(F0)
        np.testing.assert_array_equal(idx, z[f"idx_{F0}"], err_msg=f"F0={F0}")
        n = len(idx)
        assert n == O.subsample_count(F0)
        np.testing.assert_array_equal(O.fine_frame_indices(n), z[f"fine_{F0}"], err_msg=f"F0={F0}")
        np.testing.assert_array_equal(np.array(O.uniform_segment_variant(n, 32)), z[f"bounds_{F0}"])


def test_g6_glue_end_to_end():
    z, m = load_golden("g6_glue.npz")
    D, side = m["D"], m["side"]
    cfg = O.PathConfig(hidden=D, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    emb = np.zeros((48900, D), dtype=np.float32)
    emb[m["emb_rows"]] = O.bf16_round(O.hash_normal_like((len(m["emb_rows"]), D), m["embseed"], 0.02))
    # 2x2 bilinear pool (llava_arch.py:277-297)
    feats8 = O.bf16_round(O.hash_normal_like((8, side * side, D), int(z["pool_in_seed"])))
    np.testing.assert_allclose(O.bilinear_pool(feats8[:2], side), z["pooled_2"], rtol=0, atol=1e-5)  # 1-2 ulp: fp32 lerp association
    for F0 in (8, 70, 330):
        feats = O.bf16_round(O.hash_normal_like((F0, side * side, D), m["featseed0"] + F0))
        idx = O.subsample_indices(F0)
        x = O.bilinear_pool(feats[idx], side)
        toks, parts = O.video_tokens(x, idx, cfg, w, emb, "fp32", return_parts=True)
        ids = np.array([[11, 22, O.IMAGE_TOKEN_INDEX, 33, 44]])
        labels = np.array([[-100, -100, -100, 33, 44]])
        e, lab, msk, pid = O.splice(ids, labels, np.ones_like(ids), toks, emb, max_len=32768)
        n = len(parts["memory"])
        assert e.shape[1] == int(z[f"rows_{F0}"]) == 10 + n * 1568 + 1 + 9 + min(32, len(idx)) * 196 + 1 + 4
        step = m["rowsteps"][str(F0)]
        assert O.rel_l2(e[0, ::step], z[f"embeds_{F0}"]) < 2 * TOL, F0
        np.testing.assert_allclose(e[0].astype(np.float64).sum(axis=1), z[f"rowsum_{F0}"], atol=2e-3)
        np.testing.assert_array_equal(lab[0, -8:], z[f"labels_{F0}"])
        assert msk.all() == bool(z[f"mask_all_{F0}"])
        assert O.rel_l2(parts["frame_scores"][-1], z[f"scores_{F0}"][0]) < TOL


@pytest.mark.parametrize("tag,M,F,steps", [("m8f32", 8, 32, 3), ("m64f8", 64, 8, 1), ("m64f32", 64, 32, 2)])
def test_g7_fullsize_samples(tag, M, F, steps):
    z, m = load_golden("g7_fullsize.npz")
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=M, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    rm = O.RecurrentMemory(cfg, w, "fp32")
    rm.reset()
    for t in range(steps):
        seg = O.bf16_round(O.hash_normal_like((F, 196, 1024), m["segseed0"] + t))
        cache, scores = rm.step(seg)
        mem = cache[-1].reshape(-1)
        assert O.rel_l2(mem[::m["stride"]], z[f"{tag}_s{t}_sample"]) < 2 * TOL
        assert abs(np.linalg.norm(mem.astype(np.float64)) / float(z[f"{tag}_s{t}_norm"]) - 1) < 1e-5
        assert O.rel_l2(scores[-1], z[f"{tag}_s{t}_scores"]) < 2 * TOL


def test_g7_fifo_wrap_fullsize_samples():
    """The FIFO-wrapping chain at full width (checkpoint shape: 8 memory tokens, D = 1024; 13 steps of 1-2 frames, cap 10:
    three evictions, MemoryController.py:152-154) - the oracle's fp32 mode against the reference's own fp32 run, every step,
    and the whole FIFO after the last one (the oldest surviving entry is step 3's memory)."""
    z, m = load_golden("g7_fifo_fullsize.npz")
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    rm = O.RecurrentMemory(cfg, w, "fp32")
    rm.reset()
    for t, F in enumerate(m["frames"]):
        seg = O.bf16_round(O.hash_normal_like((F, 196, 1024), m["segseed0"] + t))
        cache, scores = rm.step(seg)
        assert len(cache) == min(t + 1, 10)
        mem = cache[-1].reshape(-1)
        assert O.rel_l2(mem[::m["stride"]], z[f"s{t}_sample"]) < 2 * TOL, t
        assert abs(np.linalg.norm(mem.astype(np.float64)) / float(z[f"s{t}_norm"]) - 1) < 1e-5
        assert O.rel_l2(scores[-1], z[f"s{t}_scores"]) < 2 * TOL
    got = np.stack([c.reshape(-1)[::m["stride"]] for c in cache])
    assert got.shape == z["final_cache_samples"].shape and O.rel_l2(got, z["final_cache_samples"]) < 2 * TOL


def test_g7_wide_fullsize_samples():
    """The OneVision-7B width (hidden 3584, head_dim 448; round 4): the oracle's fp32 mode against the reference's own fp32 run
    of 3 recurrent steps (formation, 2 x evolution + formation) - the pin of the wide-head path outside D = 1024."""
    z, m = load_golden("g7_wide_fullsize.npz")
    cfg = O.PathConfig(hidden=3584, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=m["wseed"])
    rm = O.RecurrentMemory(cfg, w, "fp32")
    rm.reset()
    for t, F in enumerate(m["frames"]):
        seg = O.bf16_round(O.hash_normal_like((F, 196, 3584), m["segseed0"] + t))
        cache, scores = rm.step(seg)
        mem = cache[-1].reshape(-1)
        assert O.rel_l2(mem[::m["stride"]], z[f"s{t}_sample"]) < 2 * TOL, t
        assert abs(np.linalg.norm(mem.astype(np.float64)) / float(z[f"s{t}_norm"]) - 1) < 1e-5
        assert O.rel_l2(scores[-1], z[f"s{t}_scores"]) < 2 * TOL


def test_g9_transformer_fuser_variant_matches_reference():
    """Inactive MemoryFuser variant (MemoryFuser.py:4-30): oracle/variants.py against the imported reference class."""
    from oracle import variants as V
    z, meta = load_golden("g9_variants.npz")
    for tag, (D, N) in meta["cases"].items():
        w = V.fuser_weights(D, seed=meta["wseed"])
        x = O.bf16_round(O.hash_normal_like((2, N, D), meta["xseed"]))
        for b in range(2):
            y = V.transformer_fuser(x[b], w, heads=meta["heads"], mode="fp32", layers=meta["layers"])
            assert O.rel_l2(y[::meta["rowstride"]], z[tag + "_out"][b]) < 2e-5, tag


def test_g9_gru_encoder_variant_matches_reference():
    """Inactive TemporalGRUEncoder (bigru.py:14-75): oracle/variants.py against the imported reference class."""
    from oracle import variants as V
    z, meta = load_golden("g9_variants.npz")
    for tag, (D, H, Fn, P, pe) in meta["gru_cases"].items():
        w = V.gru_weights(D, H, seed=meta["gru_wseed"])
        x = O.bf16_round(O.hash_normal_like((Fn, P, D), meta["gru_xseed"]))
        y = V.gru_encoder(x, w, H, "fp32", use_pe=bool(pe))
        assert O.rel_l2(y, z[tag]) < 1e-5, tag


def test_g9_scene_similarity_oracle_matches_reference():
    from oracle import variants as V
    z, meta = load_golden("g9_variants.npz")
    for tag, (Tn, slen, num, k, alpha) in meta["seg_cases"].items():
        feats = V.scene_features(Tn, meta["seg_P"], meta["seg_D"], slen, meta["seg_seed"])
        sims = V.adjacent_cosine(V.frame_means(feats))
        np.testing.assert_allclose(sims, z[tag + "_sims"], atol=2e-6)


def test_query_blocked_attention_equals_unblocked(monkeypatch):
    """The oracle processes very large attentions (BASELINE configs[4]) in blocks of 3136 queries: same values as the
    one-shot evaluation, in both arithmetic modes, column sums included."""
    from oracle import memory_path as O
    R, Lk, H, d = 3136 * 2 + 40, 192, 2, 64
    Q = O.bf16_round(O.hash_normal_like((R, H * d), 1))
    K = O.bf16_round(O.hash_normal_like((Lk, H * d), 2))
    V = O.bf16_round(O.hash_normal_like((Lk, H * d), 3))
    for mode in ("fp32", "bf16"):
        ref = O._attention_heads(Q, K, V, H, mode, True, False, None, None, R)
        monkeypatch.setattr(O, "ROW_BLOCK_ELEMS", 1)
        got = O.attention_heads(Q, K, V, H, mode, want_colsum=True)
        monkeypatch.setattr(O, "LAZY_SCORE_ELEMS", 1)          # + scores produced per key chunk instead of as one matrix
        lazy = O.attention_heads(Q, K, V, H, mode, want_colsum=True)
        monkeypatch.undo()
        for out in (got, lazy):
            assert O.rel_l2(out[0], ref[0]) < 1e-6 and O.rel_l2(out[1], ref[1]) < 1e-6 and O.rel_l2(out[2], ref[2]) < 1e-6
