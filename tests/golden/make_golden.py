#!/usr/bin/env python3
"""Generate golden vectors by running the *reference's own Python* on CPU.

Runs ONLY in the build container (needs /root/reference, read-only).  Nothing here travels
as code to the GPU box except this script itself; its outputs (``tests/golden/*.npz``) are data:
inputs are regenerated from the counter-based generator in ``oracle/memory_path.py`` (seeds stored
in the files), outputs are what the reference computed.

Import recipe: single reference files are loaded in isolation with
``importlib.util.spec_from_file_location``; the glue in ``llava/model/llava_arch.py`` is imported
through an empty ``llava`` namespace module plus three compatibility names on
``transformers.modeling_utils`` (the reference pins transformers 4.40-dev; the image has 5.x).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import importlib.util
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("HF_HUB_OFFLINE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import memory_path as O  # noqa: E402  (input/weight generator shared with the tests)

torch.manual_seed(0)
torch.set_grad_enabled(False)


def load_ref_file(name):
    path = f"{REF}/llava/model/memory_module/{name}.py"
    spec = importlib.util.spec_from_file_location(f"ref_{name}", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


MC = load_ref_file("MemoryController")
PEm = load_ref_file("position_encoding")
SEG = load_ref_file("segment")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def ref_projector(cfg: O.PathConfig, w, dtype=torch.float32):
    c = MC.Config()
    c.mm_hidden_size = cfg.hidden
    c.mm_intermediate_size = cfg.inter
    c.mm_num_attention_heads = cfg.heads
    c.num_memory_tokens = cfg.mem_tokens
    c.patch_size = cfg.patches
    c.depth = cfg.depth
    c.mm_layer_norm_eps = cfg.eps
    c.mm_dtype = torch.float32
    m = MC.TransformerProjector(c).eval()
    pfx = "recurrent_memory_transformer."
    sd = {k[len(pfx):]: T(v) for k, v in w.items() if k.startswith(pfx)}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m.to(dtype)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


def meta(**kw):
    kw["torch"] = torch.__version__
    return np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)


# --------------------------------------------------------------------------- G1 / G2
def g1_g2():
    cfg = O.PathConfig(hidden=128, heads=8, mem_tokens=2, depth=1)
    w = O.make_weights(cfg, seed=11)
    m = ref_projector(cfg, w)
    R, S = 392, 588
    xq = O.bf16_round(O.hash_normal_like((R, 128), 101))
    xkv = O.bf16_round(O.hash_normal_like((S, 128), 102))
    att = m.layers[0].memory_segment_fusion_attention
    out, probs = att(T(xq)[None], kv_hidden_states=T(xkv)[None])
    colsum = probs.sum(dim=1).sum(dim=1).squeeze(0)
    lay_out, lay_probs = m.layers[0](T(xq)[None], T(xkv)[None])
    save("g1_attention.npz", meta=meta(hidden=128, heads=8, R=R, S=S, wseed=11, qseed=101, kvseed=102, mem_tokens=2,
                                        depth=1),
         out=out[0].numpy(), colsum=colsum.numpy(), probs_h0_rows=probs[0, 0, :4].numpy(),
         layer_out=lay_out[0].numpy())


# --------------------------------------------------------------------------- G3
def g3():
    # NB: the reference's dead statistics code (MemoryController.py:109, reshape(8,-1,8)) raises for
    # any num_memory_tokens that is not a multiple of 8 once the cache is non-empty -> M=8 here.
    cfg = O.PathConfig(hidden=64, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=31)
    m = ref_projector(cfg, w)
    m.memory_cache = []
    m.frame_attn_scores = []
    outs = {}
    F = 3
    for t in range(4):
        seg = O.bf16_round(O.hash_normal_like((F, 196, 64), 300 + t))
        cache, scores = m(T(seg))
        full = cache[-1].numpy()
        outs[f"mem{t}"] = full.copy() if t == 3 else full[:, ::4, :].copy()
        outs[f"mem{t}_sum"] = np.array(full.astype(np.float64).sum())
        outs[f"score{t}"] = scores[-1].numpy().copy()
        assert len(cache) == t + 1
    save("g3_recurrent.npz", meta=meta(hidden=64, heads=8, mem_tokens=8, depth=2, F=F, steps=4, wseed=31, segseed0=300, rowstride_first3=4),
         **outs)

    # FIFO eviction: 12 steps, cap 10, ragged chunk sizes (1 or 2 frames)
    cfg = O.PathConfig(hidden=32, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=32)
    m = ref_projector(cfg, w)
    m.memory_cache = []
    m.frame_attn_scores = []
    fr = [1, 2, 1, 1, 2, 1, 1, 1, 2, 1, 1, 2]
    for t, f in enumerate(fr):
        seg = O.bf16_round(O.hash_normal_like((f, 196, 32), 400 + t))
        cache, scores = m(T(seg))
    assert len(cache) == 10
    save("g3_fifo.npz", meta=meta(hidden=32, heads=8, mem_tokens=8, depth=2, frames=fr, wseed=32, segseed0=400, rowstride=8),
         cache=torch.stack(cache).numpy()[:, :, ::8, :].copy(),
         cache_sums=torch.stack(cache).double().sum(dim=(1, 2, 3)).numpy(), scores_last=scores[-1].numpy(), n_scores=np.array(len(scores)))


# --------------------------------------------------------------------------- G4
def g4():
    D = 64
    pe = PEm.TemporalPositionalEncoding(max_frames=600, embed_dim=D, learnable=False)
    table = pe.frame_embed.numpy()
    pe1024 = PEm.TemporalPositionalEncoding(max_frames=600, embed_dim=1024, learnable=False).frame_embed.numpy()
    x = O.hash_normal_like((7, 196, D), 41)
    idx = np.array([0, 3, 17, 100, 333, 598, 599], dtype=np.int64)
    y = pe(T(x), T(idx)).numpy()
    ydef = pe(T(x)).numpy()
    errs = {}
    for name, bad in (("too_big", [0, 600]), ("negative", [-1, 5])):
        try:
            pe(T(x[:2]), T(np.array(bad, dtype=np.int64)))
            errs[name] = "no error"
        except ValueError as e:
            errs[name] = str(e)
    try:
        pe(T(x[0]))
        errs["rank2"] = "no error"
    except ValueError as e:
        errs["rank2"] = str(e)
    save("g4_pe.npz", meta=meta(D=D, xseed=41, errors=errs), table=table, idx=idx, y=y, ydef=ydef,
         table1024_rows=pe1024[[0, 1, 2, 299, 599]], table1024_sum=np.array(pe1024.astype(np.float64).sum()))


# --------------------------------------------------------------------------- glue import
def import_glue():
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for n in ("apply_chunking_to_forward", "find_pruneable_heads_and_indices", "prune_linear_layer"):
        if not hasattr(mu, n):
            if hasattr(pu, n):
                setattr(mu, n, getattr(pu, n))
            else:
                def _stub(*a, _n=n, **k):
                    raise NotImplementedError(_n)
                setattr(mu, n, _stub)
    pkg = types.ModuleType("llava")
    pkg.__path__ = [f"{REF}/llava"]
    sys.modules["llava"] = pkg
    from llava.model.llava_arch import LlavaMetaModel, LlavaMetaForCausalLM
    return LlavaMetaModel, LlavaMetaForCausalLM


# --------------------------------------------------------------------------- G5 / G6
def g5_g6():
    LlavaMetaModel, LlavaMetaForCausalLM = import_glue()
    import torch.nn as nn
    D = 32
    VOCAB = 48900
    SIDE = 27

    class Cfg:
        hidden_size = D
        mm_patch_merge_type = "spatial_unpad"
        mm_newline_position = "one_token"
        mm_spatial_pool_mode = "bilinear"
        image_aspect_ratio = "anyres_max_9"
        tokenizer_model_max_length = 32768
        tokenizer_padding_side = "right"

    class FakeTower(nn.Module):
        num_patches_per_side = SIDE

        def __init__(self):
            super().__init__()
            self.table = None

        def forward(self, images):  # images [F,1,1,1] holding the original frame id
            ids = images.reshape(-1).long()
            return self.table[ids]

    class TinyBase(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.embed_tokens = nn.Embedding(VOCAB, D)

        @property
        def device(self):
            return torch.device("cpu")

        @property
        def dtype(self):
            return torch.float32

    class TinyModel(LlavaMetaModel, TinyBase):
        pass

    class TinyLM(LlavaMetaForCausalLM, nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            self.config = Cfg()
            self.model = TinyModel(self.config)
            self.model.vision_tower = FakeTower()
            self.model.mm_projector = nn.Identity()
            self.model.image_newline = nn.Parameter(torch.zeros(D))

        def get_model(self):
            return self.model

        @property
        def device(self):
            return torch.device("cpu")

    cfg = O.PathConfig(hidden=D, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=61)
    lm = TinyLM().eval().float()  # Config.mm_dtype=fp16 is only a creation dtype (SURVEY App.A); run in fp32
    sd = lm.model.state_dict()
    for k, v in w.items():
        if k in sd:
            sd[k].copy_(T(v))
    # embed table: only the prompt + text ids are non-zero (kept tiny & reproducible)
    text_ids = [151644, 8948]  # replaced below: keep ids < VOCAB
    text_ids = [11, 22, 33, 44]
    rows = sorted(set(O.MEM_PROMPT_IDS + O.FRAME_PROMPT_IDS + text_ids))
    emb = np.zeros((VOCAB, D), dtype=np.float32)
    emb[rows] = O.bf16_round(O.hash_normal_like((len(rows), D), 62, 0.02))
    lm.model.embed_tokens.weight.copy_(T(emb))
    lm.model.image_newline.copy_(T(w["image_newline"]))

    index_cases = {}
    for F0 in (1, 8, 31, 32, 33, 63, 64, 65, 95, 96, 100, 127, 128, 200, 330, 599):
        n = F0 if F0 < 32 else max(64, (F0 // 32) * 32)
        idx = torch.linspace(0, F0 - 1, steps=n).long().numpy()
        ns = min(32, n)
        fine = torch.clamp(torch.round(torch.linspace(0, n - 1, steps=ns)).long(), 0, n - 1).numpy()
        bounds = np.array(SEG.uniform_segment_variant(torch.zeros(n, 1), d=32))
        index_cases[f"idx_{F0}"] = idx
        index_cases[f"fine_{F0}"] = fine
        index_cases[f"bounds_{F0}"] = bounds

    e2e = {}
    for F0, rowstep in ((8, 1), (70, 13), (330, 29)):
        feats = O.bf16_round(O.hash_normal_like((F0, SIDE * SIDE, D), 600 + F0))
        lm.model.vision_tower.table = T(feats)
        images = [torch.arange(F0, dtype=torch.float32).reshape(F0, 1, 1, 1)]
        input_ids = torch.tensor([[11, 22, -200, 33, 44]])
        labels = torch.tensor([[-100, -100, -100, 33, 44]])
        am = torch.ones_like(input_ids)
        lm.get_model().recurrent_memory_transformer.frame_attn_scores = []
        out = lm.prepare_inputs_labels_for_multimodal(input_ids, None, am, None, labels, images, modalities=["video"])
        _, pos, mask, _, embeds, labs = out
        embeds = embeds[0].numpy()
        e2e[f"rows_{F0}"] = np.array(embeds.shape[0])
        e2e[f"embeds_{F0}"] = embeds[::rowstep].copy()
        e2e[f"rowsum_{F0}"] = embeds.astype(np.float64).sum(axis=1)
        e2e[f"labels_{F0}"] = labs[0].numpy()[-8:]
        e2e[f"mask_all_{F0}"] = np.array(bool(mask.all()))
        e2e[f"scores_{F0}"] = torch.stack(lm.get_model().recurrent_memory_transformer.frame_attn_scores[-1:]).numpy()
        if F0 == 8:
            pooled = lm.get_2dPool(T(feats[:2]))
            e2e["pool_in_seed"] = np.array(600 + F0)
            e2e["pooled_2"] = pooled.numpy()
    # state-dict contract of the memory modules (names, shapes) as the reference builds them
    sd_names = {k: list(v.shape) for k, v in lm.model.state_dict().items()
                if k.split(".")[0] in ("recurrent_memory_transformer", "memory_fuser", "positional_encoding",
                                       "token_type_embedding")}
    with open(os.path.join(HERE, "g6_statedict.json"), "w") as f:
        json.dump({"hidden": D, "keys": sd_names}, f, indent=0, sort_keys=True)
    save("g5_index.npz", meta=meta(note="torch.linspace/.long()/round index math"), **index_cases)
    save("g6_glue.npz", meta=meta(D=D, side=SIDE, wseed=61, embseed=62, emb_rows=rows, text_ids=text_ids,
                                   featseed0=600, rowsteps={"8": 1, "70": 13, "330": 29}), **e2e)


# --------------------------------------------------------------------------- G7
def g7():
    res = {}
    # m64f32 = the exact bench.py workload (BASELINE.json configs[1]): M = 64, two 32-frame chunks (formation, then
    # evolution over the FIFO + formation)
    for tag, M, F, steps in (("m8f32", 8, 32, 3), ("m64f8", 64, 8, 1), ("m64f32", 64, 32, 2)):
        cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=M, depth=2)
        w = O.make_weights(cfg, seed=71)
        m = ref_projector(cfg, w)
        m.memory_cache = []
        m.frame_attn_scores = []
        for t in range(steps):
            seg = O.bf16_round(O.hash_normal_like((F, 196, 1024), 700 + t))
            cache, scores = m(T(seg))
            mem = cache[-1].numpy().reshape(-1)
            res[f"{tag}_s{t}_sample"] = mem[::997].copy()
            res[f"{tag}_s{t}_norm"] = np.array(np.linalg.norm(mem.astype(np.float64)))
            res[f"{tag}_s{t}_sum"] = np.array(mem.astype(np.float64).sum())
            res[f"{tag}_s{t}_scores"] = scores[-1].numpy().copy()
        # reference bf16 run of the same thing: documents the reference's own bf16-vs-fp32 envelope
        mb = ref_projector(cfg, w, torch.bfloat16)
        mb.memory_cache = []
        mb.frame_attn_scores = []
        for t in range(steps):
            seg = O.bf16_round(O.hash_normal_like((F, 196, 1024), 700 + t))
            cache, scores = mb(T(seg).to(torch.bfloat16))
            mem = cache[-1].float().numpy().reshape(-1)
            res[f"{tag}_s{t}_sample_refbf16"] = mem[::997].copy()
    save("g7_fullsize.npz", meta=meta(hidden=1024, wseed=71, segseed0=700, stride=997), **res)


G7_FIFO_FRAMES = (2, 1, 2, 2, 1, 2, 1, 1, 2, 2, 1, 2, 2)      # 13 steps: the FIFO (cap 10) is full after 10, evicts 3 times
G7_WIDE_FRAMES = (2, 1, 2)                                    # OneVision-7B width: formation, 2 x (evolution + formation)


def g7_fifo():
    """A FIFO-wrapping chain at FULL width the HIP path can run (round 3): the checkpoint shape (8 memory tokens, D = 1024,
    8 heads, depth 2), 13 steps of 1-2 frames - from step 10 on every append evicts the oldest memory
    (MemoryController.py:152-154) and the evolution attends over all 10 cached memories.  Per step: strided samples + norm +
    sum of the new memory and the frame scores of the reference's fp32 run, and the samples of its bf16 run."""
    res = {}
    cfg = O.PathConfig(hidden=1024, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=72)
    for dtype, sfx in ((torch.float32, ""), (torch.bfloat16, "_refbf16")):
        m = ref_projector(cfg, w, dtype)
        m.memory_cache = []
        m.frame_attn_scores = []
        for t, F in enumerate(G7_FIFO_FRAMES):
            seg = O.bf16_round(O.hash_normal_like((F, 196, 1024), 7200 + t))
            cache, scores = m(T(seg).to(dtype))
            assert len(cache) == min(t + 1, 10)
            mem = cache[-1].float().numpy().reshape(-1)
            res[f"s{t}_sample{sfx}"] = mem[::499].copy()
            if not sfx:
                res[f"s{t}_norm"] = np.array(np.linalg.norm(mem.astype(np.float64)))
                res[f"s{t}_sum"] = np.array(mem.astype(np.float64).sum())
                res[f"s{t}_scores"] = scores[-1].numpy().copy()
                if t == len(G7_FIFO_FRAMES) - 1:          # the whole FIFO after the last step: oldest entry = step 3's memory
                    res["final_cache_samples"] = np.stack([c.numpy().reshape(-1)[::499] for c in cache])
    save("g7_fifo_fullsize.npz", meta=meta(hidden=1024, wseed=72, segseed0=7200, stride=499, frames=list(G7_FIFO_FRAMES)), **res)


def g7_wide():
    """The OneVision-7B width (round 4): hidden 3584, 8 heads -> head_dim 448 (llava_arch.py:117-122 with the 7B backbone's
    hidden size), the checkpoint's 8 memory tokens, depth 2; 3 recurrent steps of 2 / 1 / 2 frames (formation, then evolution +
    formation twice).  Per step: strided samples + norm of the new memory and the frame scores of the reference's fp32 run, the
    samples of its bf16 run.  Pins the wide-head kernels (attention_hd.hip) to the REFERENCE, not just to the emulation oracle."""
    res = {}
    cfg = O.PathConfig(hidden=3584, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=73)
    for dtype, sfx in ((torch.float32, ""), (torch.bfloat16, "_refbf16")):
        m = ref_projector(cfg, w, dtype)
        m.memory_cache = []
        m.frame_attn_scores = []
        for t, F in enumerate(G7_WIDE_FRAMES):
            seg = O.bf16_round(O.hash_normal_like((F, 196, 3584), 7300 + t))
            cache, scores = m(T(seg).to(dtype))
            mem = cache[-1].float().numpy().reshape(-1)
            res[f"s{t}_sample{sfx}"] = mem[::997].copy()
            if not sfx:
                res[f"s{t}_norm"] = np.array(np.linalg.norm(mem.astype(np.float64)))
                res[f"s{t}_scores"] = scores[-1].numpy().copy()
    save("g7_wide_fullsize.npz", meta=meta(hidden=3584, wseed=73, segseed0=7300, stride=997, frames=list(G7_WIDE_FRAMES)), **res)


# --------------------------------------------------------------------------- G8 (gradients)
G8_CASES = {"d256": (256, 7), "d1024": (1024, 61)}      # hidden (8 heads: the reference's dead reshape needs H = M = 8), stride
G8_WIDE = {"d3584": (3584, 1999)}                       # the OneVision-7B width (heads of 448; `python make_golden.py g8wide`, ~1 h of CPU)


def g8_case(dtype, hidden, stride, small_full=0):
    """3 recurrent steps of the reference TransformerProjector under autograd (BPTT through the un-detached
    memory_cache, MemoryController.py:125-127,152), loss = sum_t <cache[t], G_t>; gradients of every parameter."""
    cfg = O.PathConfig(hidden=hidden, heads=8, mem_tokens=8, depth=2)
    w = O.make_weights(cfg, seed=81)
    m = ref_projector(cfg, w, dtype).train()
    m.memory_cache = []
    m.frame_attn_scores = []
    frames = [2, 1, 2]
    with torch.enable_grad():
        for t, f in enumerate(frames):
            seg = O.bf16_round(O.hash_normal_like((f, 196, hidden), 800 + t))
            cache, _ = m(T(seg).to(dtype))
        loss = 0.0
        for t, c in enumerate(cache):
            g = O.bf16_round(O.hash_normal_like(tuple(c.shape), 850 + t, 0.05))
            loss = loss + (c.float() * T(g)).sum()
        loss.backward()
    out = {"loss": np.array(float(loss))}
    for name, p in m.named_parameters():
        g = p.grad.float().numpy().reshape(-1)
        # (wide case: parameters of <= small_full elements - biases, LayerNorm affines - are stored whole: 2 samples of a
        # 3584-vector make no envelope)
        out["g_" + name + "_sample"] = g[::(1 if g.size <= small_full else stride)].copy()
        out["g_" + name + "_norm"] = np.array(np.linalg.norm(g.astype(np.float64)))
    return cfg, frames, out


def g8(cases=None):
    small_full = 65536 if cases is G8_WIDE else 0
    for tag, (hidden, stride) in (cases or G8_CASES).items():
        cfg, frames, ref = g8_case(torch.float32, hidden, stride, small_full)
        _, _, bf = g8_case(torch.bfloat16, hidden, stride, small_full)
        # the reference's own bf16-vs-fp32 distance per parameter gradient (on the stored samples): the envelope the
        # 16-bit HIP backward is judged against
        env = {}
        for k in ref:
            if k.endswith("_sample"):
                env["env_" + k[2:-7]] = np.array(O.rel_l2(bf[k], ref[k]))
        save(f"g8_grads_{tag}.npz", meta=meta(hidden=hidden, heads=8, mem_tokens=8, depth=2, frames=frames, wseed=81,
                                              segseed0=800, gseed0=850, gstd=0.05, stride=stride, small_full=small_full), **ref, **env)
        print(tag, "reference bf16-vs-fp32 gradient envelope: max", max(float(v) for v in env.values()),
              "median", float(np.median([float(v) for v in env.values()])))


# --------------------------------------------------------------------------- G9 (inactive variants)
def g9():
    from oracle import variants as V
    MF = load_ref_file("MemoryFuser")
    out = {}
    cases = {"d512": (512, 150), "d1024": (1024, 300), "d896": (896, 70)}
    for tag, (D, N) in cases.items():
        w = V.fuser_weights(D, seed=91)
        m = MF.MemoryFuser(D, num_layers=2, num_heads=4, device="cpu").eval()
        m.load_state_dict({k: T(v) for k, v in w.items()}, strict=True)
        x = O.bf16_round(O.hash_normal_like((2, N, D), 910))
        y = m(T(x))
        out[tag + "_out"] = y.numpy()[:, ::3, :].copy()
        out[tag + "_sum"] = np.array(y.double().sum().item())
    # TemporalGRUEncoder (bigru.py)
    BG = load_ref_file("bigru")
    gru_cases = {"gru896": (896, 448, 20, 6, False), "gru896pe": (896, 448, 9, 4, True), "gru1024": (1024, 512, 33, 3, False)}
    for tag, (D, H, Fn, P, pe) in gru_cases.items():
        w = V.gru_weights(D, H, seed=95)
        m = BG.TemporalGRUEncoder(input_dim=D, hidden_size=H, use_positional_encoding=pe).eval()
        sd = {k: T(v) for k, v in w.items()}
        if pe:
            sd["temporal_pe"] = m.temporal_pe
        m.load_state_dict(sd, strict=True)
        x = O.bf16_round(O.hash_normal_like((Fn, P, D), 950))
        out[tag] = m(T(x)).numpy()
    # scene segmentation / sampling (segment.py)
    seg_cases = {"s40": (40, 10, 32, None, 0.3), "s120": (120, 15, 32, None, 0.3), "s100many": (100, 2, 32, None, 0.3),
                 "s64k": (64, 8, 16, 5, 0.5), "s33": (33, 33, 32, None, 0.3), "s200": (200, 7, 32, None, 0.0)}
    for tag, (Tn, slen, num, k, alpha) in seg_cases.items():
        feats = T(V.scene_features(Tn, 4, 64, slen, 970))
        means = feats.mean(dim=1)
        sims = torch.cosine_similarity(means[:-1], means[1:], eps=1e-2)
        b, depth = SEG.segment(means, alpha=alpha, k=k)
        torch.manual_seed(4242)
        idx = SEG.sample_scenes_priority(feats, sample_num=num, alpha=alpha, k=k)
        out[tag + "_sims"] = sims.numpy()
        out[tag + "_depth"] = depth.numpy()
        out[tag + "_bounds"] = np.array(b, dtype=np.int64)
        out[tag + "_idx"] = np.array(idx, dtype=np.int64)
    import contextlib, io
    adj = {"a300": (300, 23, None, 0.5, 32, 64), "a90": (90, 9, None, 0.3, 16, 24), "a500k": (500, 40, 6, 0.5, 32, 64),
           "a40": (40, 50, None, 0.5, 32, 64)}
    for tag, (Tn, slen, k, alpha, mind, maxd) in adj.items():
        feats = T(V.scene_features(Tn, 4, 64, slen, 980)).mean(dim=1)
        with contextlib.redirect_stdout(io.StringIO()):                # the reference prints its boundaries
            b = SEG.adjusted_segment(feats, alpha=alpha, k=k, min_distance=mind, max_distance=maxd)
        out[tag + "_sims"] = torch.cosine_similarity(feats[:-1], feats[1:]).numpy()
        out[tag + "_adj"] = np.array(b, dtype=np.int64)
    out["uniform_segment"] = np.array([len(SEG.uniform_segment(torch.zeros(t, 1), d=32)) for t in range(1, 140)], dtype=np.int64)
    out["uniform_segment_70"] = np.array(SEG.uniform_segment(torch.zeros(70, 1), d=32), dtype=np.int64)
    save("g9_variants.npz", meta=meta(cases={k: list(v) for k, v in cases.items()}, wseed=91, xseed=910, rowstride=3,
                                      heads=4, layers=2, gru_cases={k: list(v) for k, v in gru_cases.items()},
                                      gru_wseed=95, gru_xseed=950, seg_cases={k: list(v) for k, v in seg_cases.items()},
                                      seg_seed=970, seg_rng=4242, seg_P=4, seg_D=64,
                                      adj_cases={k_: [x if x is not None else -1 for x in v] for k_, v in adj.items()},
                                      adj_seed=980), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g3", "g4", "g56", "g7", "g7fifo", "g7wide", "g8", "g9"]
    if "g9" in which:
        g9()
    if "g8" in which:
        g8()
    if "g8wide" in which:
        g8(G8_WIDE)
    if "g1" in which:
        g1_g2()
    if "g3" in which:
        g3()
    if "g4" in which:
        g4()
    if "g56" in which:
        g5_g6()
    if "g7" in which:
        g7()
    if "g7wide" in which:
        g7_wide()
    if "g7fifo" in which:
        g7_fifo()
