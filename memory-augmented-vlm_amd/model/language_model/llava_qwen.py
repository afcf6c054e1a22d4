"""HF wrapper that routes a Qwen2 causal LM through the MI355X memory path.

Drop-in for llava/model/language_model/llava_qwen.py of the reference (:34-156): same class names
(`LlavaQwenConfig`, `LlavaQwenModel`, `LlavaQwenForCausalLM`), the same `forward` / `generate` /
`prepare_inputs_for_generation` signatures and the same `AutoConfig` / `AutoModelForCausalLM` registration under
model_type "llava_qwen".  The backbone is the unchanged `transformers.Qwen2ForCausalLM` on PyTorch-ROCm; the only thing
this file adds is WHERE the multimodal hand-off happens:

  forward(images=...)    training / teacher forcing: `prepare_inputs_labels_for_multimodal` (model/llava_arch.py, the HIP
                         memory path) turns (input_ids, images) into (inputs_embeds, labels, mask, position ids), then the
                         stock Qwen2 forward runs on the embeddings                                  (llava_qwen.py:80-114)
  generate(images=...)   inference: the memory path runs ONCE, for the prefill; decode steps arrive with one new token
                         and no images and take the early exit of the glue (llava_arch.py:392-394) (llava_qwen.py:116-136)

Differences from the reference, all forced by the installed transformers (5.x; the reference pins a 4.40 dev commit):
  * `config.rope_scaling = None` (llava_qwen.py:51) is applied only to configs that still carry the legacy field; on
    transformers 5 `rope_scaling` aliases `rope_parameters` and clearing it breaks the rotary embedding.
  * `output_attentions` / `output_hidden_states` / `return_dict` / `cache_position` are forwarded only when set: the 5.x
    `Qwen2ForCausalLM.forward` takes them through **kwargs.
The vision tower and `mm_projector` are the backbone's (out of scope); attach them with
`LlavaQwenModel.attach_vision_modules` (or assign the attributes, as the reference's `initialize_vision_modules` does).
"""
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn
from transformers import AutoConfig, AutoModelForCausalLM, Qwen2Config, Qwen2ForCausalLM, Qwen2Model
from transformers.modeling_outputs import CausalLMOutputWithPast

from ..llava_arch import LlavaMetaForCausalLM, LlavaMetaModel


class LlavaQwenConfig(Qwen2Config):
    """Qwen2Config under model_type "llava_qwen" (llava_qwen.py:34-35).  Fields the memory glue reads, with the
    reference's defaults: mm_patch_merge_type "spatial_unpad", mm_newline_position "one_token", mm_spatial_pool_mode
    "bilinear", tokenizer_model_max_length, tokenizer_padding_side; optional extras of this build: num_memory_tokens
    (reference constant 8), memory_cache_cap (10), memory_max_frames (600)."""
    model_type = "llava_qwen"


class LlavaQwenModel(LlavaMetaModel, Qwen2Model):
    config_class = LlavaQwenConfig

    def __init__(self, config: Qwen2Config):
        super(LlavaQwenModel, self).__init__(config)       # Qwen2Model first, then the four memory sub-modules
        if "unpad" in getattr(config, "mm_patch_merge_type", "") and not hasattr(self, "image_newline"):
            self.image_newline = nn.Parameter(torch.zeros(config.hidden_size))          # llava_arch.py:113-114

    def attach_vision_modules(self, vision_tower, mm_projector, image_newline: Optional[torch.Tensor] = None):
        """Hand the backbone's vision modules to the glue: `vision_tower(images[F,3,h,w]) -> [F, side*side, C]` with a
        `num_patches_per_side` attribute, `mm_projector` C -> hidden (e.g. multimodal_projector.build_vision_projector)."""
        self.vision_tower = vision_tower
        self.mm_projector = mm_projector
        if image_newline is not None:
            self.image_newline = nn.Parameter(image_newline.detach().clone())
        return self


class LlavaQwenForCausalLM(Qwen2ForCausalLM, LlavaMetaForCausalLM):
    config_class = LlavaQwenConfig

    def __init__(self, config):
        Qwen2ForCausalLM.__init__(self, config)
        config.model_type = "llava_qwen"
        if "rope_scaling" in config.__dict__ and not hasattr(config, "rope_parameters"):
            config.rope_scaling = None                       # llava_qwen.py:51 (legacy configs only, see module doc)
        self.model = LlavaQwenModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.post_init()
        self.multimodal_prefills = 0     # how many times the memory path ran (tests: decode steps must not add to it)

    def get_model(self):
        return self.model

    def prepare_inputs_labels_for_multimodal(self, *args, **kwargs):
        out = LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal(self, *args, **kwargs)
        if out[4] is not None:
            self.multimodal_prefills += 1
        return out

    def forward(
        self,
        input_ids: torch.LongTensor = None,
        attention_mask: Optional[torch.Tensor] = None,
        position_ids: Optional[torch.LongTensor] = None,
        past_key_values=None,
        inputs_embeds: Optional[torch.FloatTensor] = None,
        labels: Optional[torch.LongTensor] = None,
        use_cache: Optional[bool] = None,
        output_attentions: Optional[bool] = None,
        output_hidden_states: Optional[bool] = None,
        images: Optional[torch.FloatTensor] = None,
        image_sizes: Optional[List[List[int]]] = None,
        return_dict: Optional[bool] = None,
        modalities: Optional[List[str]] = ["image"],
        dpo_forward: Optional[bool] = False,
        cache_position=None,
        **kwargs,
    ) -> Union[Tuple, CausalLMOutputWithPast]:
        if inputs_embeds is None:                                                       # llava_qwen.py:80-82
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = \
                self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values, labels,
                                                          images, modalities, image_sizes)
        extra = {k: v for k, v in (("output_attentions", output_attentions), ("output_hidden_states", output_hidden_states),
                                   ("return_dict", return_dict), ("cache_position", cache_position)) if v is not None}
        extra.update(kwargs)
        if dpo_forward:                                                                 # :85-100: logits + spliced labels
            outputs = self.model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                 past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache, **extra)
            return self.lm_head(outputs[0]), labels
        return Qwen2ForCausalLM.forward(self, input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                        past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels,
                                        use_cache=use_cache, **extra)

    @torch.no_grad()
    def generate(
        self,
        inputs: Optional[torch.Tensor] = None,
        images: Optional[torch.Tensor] = None,
        image_sizes: Optional[torch.Tensor] = None,
        modalities: Optional[List[str]] = ["image"],
        **kwargs,
    ):
        position_ids = kwargs.pop("position_ids", None)
        attention_mask = kwargs.pop("attention_mask", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")               # llava_qwen.py:127-128
        if images is not None:                                                          # :130-132: ONE pass of the memory path
            (inputs, position_ids, attention_mask, _, inputs_embeds, _) = self.prepare_inputs_labels_for_multimodal(
                inputs, position_ids, attention_mask, None, None, images, modalities, image_sizes=image_sizes)
            if inputs_embeds is None:                    # glue took its early exit (no vision tower): text-only prompt
                inputs_embeds = self.get_model().embed_tokens(inputs)
        else:
            inputs_embeds = self.get_model().embed_tokens(inputs)
        return Qwen2ForCausalLM.generate(self, position_ids=position_ids, attention_mask=attention_mask,
                                         inputs_embeds=inputs_embeds, **kwargs)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        images = kwargs.pop("images", None)                                             # llava_qwen.py:138-150
        image_sizes = kwargs.pop("image_sizes", None)
        inputs = Qwen2ForCausalLM.prepare_inputs_for_generation(self, input_ids, past_key_values=past_key_values,
                                                                inputs_embeds=inputs_embeds, **kwargs)
        if images is not None:
            inputs["images"] = images
        if image_sizes is not None:
            inputs["image_sizes"] = image_sizes
        return inputs


def _register():
    try:
        AutoConfig.register("llava_qwen", LlavaQwenConfig)                              # llava_qwen.py:155-156
        AutoModelForCausalLM.register(LlavaQwenConfig, LlavaQwenForCausalLM)
    except ValueError:
        pass        # already registered in this process (the reference's own llava package, or a re-import)


_register()
