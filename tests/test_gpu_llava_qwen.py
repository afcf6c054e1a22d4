"""-m gpu: the HF wrapper row (SURVEY.md §8b "Outer API", BASELINE.json configs[1]/[2]: "feeding Qwen2-0.5B backbone" /
"LLaVA-OneVision-7B backbone") - `LlavaQwenForCausalLM.forward / generate` over a real `transformers.Qwen2ForCausalLM`
(random init from a local config: no weights exist offline), a 64-frame video through the HIP memory path.

Checked: the `inputs_embeds` the backbone receives are BIT-IDENTICAL to the stand-alone path (pool -> video_memory_tokens
-> manual splice), labels / mask / position ids line up, `generate()` runs the memory path exactly once (decode steps take
the early exit, llava_arch.py:392-394) and agrees with a manual greedy loop over the same backbone, and a training step
(loss.backward()) reaches the memory parameters through the backbone."""
import types

import pytest
import torch
import torch.nn as nn

import memory_augmented_vlm_amd  # noqa: F401
from memory_augmented_vlm_amd.model import llava_arch as arch
from memory_augmented_vlm_amd.model.language_model.llava_qwen import LlavaQwenConfig, LlavaQwenForCausalLM
from memory_augmented_vlm_amd.model.multimodal_projector import build_vision_projector

pytestmark = pytest.mark.gpu
VOCAB = 49152          # >= the largest fixed prompt id (48876, llava_arch.py:708,714)


class PatchTower(nn.Module):
    """Stand-in for the SigLIP tower's INTERFACE (siglip_encoder.py:577-590): images [F,3,54,54] -> [F, 27*27, C], one
    linear map per 2x2x3 patch; `num_patches_per_side` as `get_2dPool` reads it (llava_arch.py:278)."""
    num_patches_per_side = 27

    def __init__(self, C):
        super().__init__()
        self.proj = nn.Linear(12, C)

    def forward(self, images):
        F = images.shape[0]
        p = images.reshape(F, 3, 27, 2, 27, 2).permute(0, 2, 4, 1, 3, 5).reshape(F, 729, 12)
        return self.proj(p)


def build(hidden, layers, heads, kv_heads, inter, mm_hidden=1152, mem_tokens=8):
    cfg = LlavaQwenConfig(vocab_size=VOCAB, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                          num_attention_heads=heads, num_key_value_heads=kv_heads, max_position_embeddings=32768,
                          mm_patch_merge_type="spatial_unpad", mm_newline_position="one_token",
                          mm_spatial_pool_mode="bilinear", tokenizer_model_max_length=32768, tokenizer_padding_side="right",
                          num_memory_tokens=mem_tokens, tie_word_embeddings=False)
    torch.manual_seed(7)
    m = LlavaQwenForCausalLM(cfg)
    pc = types.SimpleNamespace(mm_hidden_size=mm_hidden, hidden_size=hidden, mm_projector_type="mlp2x_gelu")
    m.get_model().attach_vision_modules(PatchTower(mm_hidden), build_vision_projector(pc),
                                        image_newline=torch.randn(hidden) * 0.02)
    with torch.no_grad():      # HF init leaves LayerNorm affine at identity and initial_memory at xavier: keep, perturb LN
        for mod in m.get_model().recurrent_memory_transformer.modules():
            if isinstance(mod, nn.LayerNorm):
                mod.weight.add_(torch.rand_like(mod.weight) * 0.2 - 0.1)
    return m.to("cuda").to(torch.bfloat16).eval()


def prompt(n_before=5, n_after=6, seed=3):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, VOCAB, (n_before,), generator=g)
    b = torch.randint(0, VOCAB, (n_after,), generator=g)
    ids = torch.cat([a, torch.tensor([arch.IMAGE_TOKEN_INDEX]), b])[None].cuda()
    return ids, n_before


def standalone_embeds(m, video, ids, p):
    """The same hand-off WITHOUT the wrapper: encode -> pool -> video_memory_tokens -> cat with the text embeddings."""
    model = m.get_model()
    idx = arch.sample_frame_indices(video.shape[0])
    pooled = m.get_2dPool(m.encode_images(video[idx.to(video.device)]))
    mp = model.embed_tokens(torch.tensor([arch.MEMORY_PROMPT_IDS], device="cuda"))[0]
    fp = model.embed_tokens(torch.tensor([arch.FRAME_PROMPT_IDS], device="cuda"))[0]
    toks, _ = arch.video_memory_tokens(model, pooled, idx, mp, fp, model.image_newline)
    row = ids[0]
    return torch.cat([model.embed_tokens(row[:p]), toks, model.embed_tokens(row[p + 1:])])[None]


@pytest.mark.parametrize("name,hidden,layers,heads,kv,inter,M", [("qwen2-0.5b-shape", 896, 2, 14, 2, 4864, 8),
                                                                  ("ov-7b-shape", 3584, 1, 28, 4, 18944, 8),
                                                                  ("configs1-literal-64-memory-tokens", 896, 2, 14, 2, 4864, 64)])
def test_forward_and_generate_through_real_qwen2_backbone(name, hidden, layers, heads, kv, inter, M):
    """(third case = BASELINE.json configs[1] as literally stated: 64 memory tokens feeding a Qwen2-0.5B-shape backbone - a
    31 381-token prefill)"""
    m = build(hidden, layers, heads, kv, inter, mem_tokens=M)
    torch.manual_seed(11)
    video = torch.randn(64, 3, 54, 54, device="cuda", dtype=torch.bfloat16)
    ids, p = prompt()
    labels = ids.clone()
    labels[ids == arch.IMAGE_TOKEN_INDEX] = arch.IGNORE_INDEX
    seen = {}

    def grab(_mod, args, kwargs):
        if kwargs.get("inputs_embeds") is not None and kwargs["inputs_embeds"].shape[1] > 1:
            seen["emb"] = kwargs["inputs_embeds"].detach().clone()
            seen["pos"], seen["mask"] = kwargs.get("position_ids"), kwargs.get("attention_mask")
    h = m.model.register_forward_pre_hook(grab, with_kwargs=True)
    with torch.no_grad():
        ref = standalone_embeds(m, video, ids, p)
        out = m(input_ids=ids, labels=labels, images=[video], modalities=["video"],
                attention_mask=torch.ones_like(ids), position_ids=torch.arange(ids.shape[1], device="cuda")[None])
    rows = arch.video_token_rows(64, M)
    L = ids.shape[1] - 1 + rows
    assert m.multimodal_prefills == 1
    assert seen["emb"].shape == (1, L, hidden) and torch.equal(seen["emb"], ref)         # bit for bit
    assert seen["mask"].shape == (1, L) and bool(seen["mask"].all())
    assert torch.equal(seen["pos"], torch.arange(L, device="cuda")[None])
    assert out.logits.shape[:2] == (1, L) and torch.isfinite(out.loss)
    # the spliced labels ignore the video block: the loss equals the backbone's own loss on the spliced sequence
    lab = torch.cat([labels[0, :p], torch.full((rows,), arch.IGNORE_INDEX, device="cuda"), labels[0, p + 1:]])[None]
    with torch.no_grad():
        direct = type(m).__mro__[1].forward(m, inputs_embeds=ref, labels=lab)
    assert torch.allclose(direct.loss.float(), out.loss.float(), rtol=1e-3, atol=1e-3)

    # generate: ONE pass of the memory path for the prefill, none for the decode steps
    before = m.multimodal_prefills
    with torch.no_grad():
        new = m.generate(ids, images=[video], modalities=["video"], max_new_tokens=2, do_sample=False)
    assert m.multimodal_prefills == before + 1
    assert new.shape == (1, 2)
    assert torch.equal(seen["emb"], ref)             # the prefill of generate() saw the same embeddings
    with torch.no_grad():                             # manual greedy loop over the same backbone
        o1 = type(m).__mro__[1].forward(m, inputs_embeds=ref, use_cache=True)
        t1 = o1.logits[0, -1].float().argmax()
        o2 = type(m).__mro__[1].forward(m, input_ids=t1.view(1, 1), past_key_values=o1.past_key_values, use_cache=True)
        t2 = o2.logits[0, -1].float().argmax()
    assert int(new[0, 0]) == int(t1) and int(new[0, 1]) == int(t2)
    h.remove()


def test_training_step_through_the_backbone_reaches_the_memory_parameters():
    """`forward(labels=...)` with autograd on (the reference's training entry, llava_qwen.py:80-114): the loss of the
    Qwen2 backbone back-propagates through inputs_embeds into the HIP backward of the memory path.  The training-mode
    embeddings equal the inference-mode ones: bit for bit outside the fused-memory rows, and inside them up to the one
    extra 16-bit rounding of the fuser's GELU pre-activation that the backward needs."""
    m = build(896, 2, 14, 2, 4864)
    m.train()
    for n_, p_ in m.named_parameters():               # mm_tunable_parts = recurrent_model + larimar_model (train.py:1708-1724)
        p_.requires_grad_(any(k in n_ for k in ("recurrent_memory_transformer", "memory_fuser", "token_type_embedding")))
    torch.manual_seed(12)
    video = torch.randn(64, 3, 54, 54, device="cuda", dtype=torch.bfloat16)
    ids, p = prompt()
    labels = ids.clone()
    labels[ids == arch.IMAGE_TOKEN_INDEX] = arch.IGNORE_INDEX
    seen = {}
    h = m.model.register_forward_pre_hook(lambda _m, a, k: seen.__setitem__("emb", k["inputs_embeds"].detach().clone()),
                                          with_kwargs=True)
    out = m(input_ids=ids, labels=labels, images=[video], modalities=["video"])
    out.loss.backward()
    h.remove()
    with torch.no_grad():
        m.eval()
        ref = standalone_embeds(m, video, ids, p)
    rows = arch.video_token_rows(64, 8)
    a, b = p + len(arch.MEMORY_PROMPT_IDS), p + len(arch.MEMORY_PROMPT_IDS) + 2 * 8 * 196       # fused memory rows
    assert seen["emb"].shape == ref.shape == (1, ids.shape[1] - 1 + rows, 896)
    assert torch.equal(seen["emb"][0, :a], ref[0, :a]) and torch.equal(seen["emb"][0, b:], ref[0, b:])
    d = (seen["emb"][0, a:b].float() - ref[0, a:b].float()).norm() / ref[0, a:b].float().norm()
    # (a freshly initialised fuser - N(0, 0.02) weights - has small pre-activations, where rounding them to 16 bits costs
    # the GELU output up to ~2 ulp: 3.9e-3 measured, gate = 1.5 x that; the golden-weight case of
    # test_full_token_block_gradients sits at 2e-3)
    print(f"training-mode vs inference-mode fused rows: {float(d):.2e}")
    assert float(d) < 6e-3
    got = {n_: p_.grad for n_, p_ in m.named_parameters() if p_.requires_grad}
    assert got and all(g is not None and torch.isfinite(g.float()).all() for g in got.values())
    nz = [n_ for n_, g in got.items() if float(g.float().abs().max()) > 0]
    for key in ("memory_fuser.0.weight", "token_type_embedding.weight", "recurrent_memory_transformer.initial_memory",
                "layers.1.residual.dense.weight", "memory_update_attention.q_proj.weight"):
        assert any(key in n_ for n_ in nz), key
