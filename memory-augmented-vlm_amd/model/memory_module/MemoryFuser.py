"""Inactive variant: the TransformerEncoder Memory-Fuser of
`llava/model/memory_module/MemoryFuser.py:4-30`, whose instantiation is commented out in the reference
(`llava_arch.py:137-143`; the shipped fuser is the MLP, `MemoryFuserMLP` in ../llava_arch.py).

Same constructor, parameters and state-dict keys as the reference class (the parameters live in a real
`nn.TransformerEncoder`, so `transformer_encoder.layers.N.self_attn.in_proj_weight`, `...linear1.weight`,
`...norm1.weight` etc. load unchanged); forward = the HIP kernels of this library:
packed in_proj GEMM -> 4-head self-attention (head_dim D/4 = 256 at D = 1024, 224 at D = 896: attention_hd.hip) ->
out_proj GEMM (fp32) -> residual + LayerNorm -> Linear+GELU -> Linear (fp32) -> residual + LayerNorm  (post-norm,
`norm_first=False`, exact-erf GELU, eps 1e-5: the nn.TransformerEncoderLayer defaults the reference relies on).
Forward only, dropout inactive (eval semantics): the variant is dead code in the reference and is not trained.
"""
import torch
import torch.nn as nn

from ... import _capi as capi
from ... import _ops as ops


class MemoryFuser(nn.Module):
    def __init__(self, hidden_dim, num_layers=2, num_heads=4, dropout=0.1, device="cuda"):
        super().__init__()
        self.device = device
        self.num_heads = num_heads
        self.input_proj = nn.Linear(hidden_dim, hidden_dim)
        layer = nn.TransformerEncoderLayer(d_model=hidden_dim, nhead=num_heads, dim_feedforward=hidden_dim * 4,
                                           dropout=dropout, batch_first=True, activation="gelu")
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers)
        self.output_proj = nn.Linear(hidden_dim, hidden_dim)
        hd = hidden_dim // num_heads
        if hidden_dim % num_heads or hd not in (128, 224, 256, 448):
            raise capi.MavlmError(f"MemoryFuser: head_dim {hd} has no HIP attention kernel (128, 224, 256, 448)")
        self.head_dim = hd

    def _layer(self, x, lyr):
        D = x.shape[1]
        sa = lyr.self_attn
        qkv = ops.linear(x, sa.in_proj_weight, sa.in_proj_bias.float())                    # [N, 3D]
        ctx, _ = ops.attention(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], self.num_heads, head_dim=self.head_dim,
                               wide_kernel=True)
        pre = ops.linear(ctx, sa.out_proj.weight, sa.out_proj.bias.float(), capi.EPI_F32)
        x = ops.layernorm(pre, lyr.norm1.weight.float(), lyr.norm1.bias.float(), lyr.norm1.eps, x.dtype, residual=x)
        h = ops.linear(x, lyr.linear1.weight, lyr.linear1.bias.float(), capi.EPI_GELU)
        pre = ops.linear(h, lyr.linear2.weight, lyr.linear2.bias.float(), capi.EPI_F32)
        return ops.layernorm(pre, lyr.norm2.weight.float(), lyr.norm2.bias.float(), lyr.norm2.eps, x.dtype, residual=x)

    @torch.no_grad()
    def forward(self, memory_tokens):
        """memory_tokens: [batch, num_segments, hidden] (MemoryFuser.py:22-30)."""
        if memory_tokens.dim() != 3:
            raise capi.MavlmError("MemoryFuser expects [batch, num_segments, hidden]")
        if not memory_tokens.is_cuda:
            raise capi.MavlmError("MemoryFuser: input is not on a GPU; the HIP path has no CPU fallback")
        outs = []
        for b in range(memory_tokens.shape[0]):                  # sequences attend within themselves only
            x = memory_tokens[b].contiguous()
            x = ops.linear(x, self.input_proj.weight, self.input_proj.bias.float())
            for lyr in self.transformer_encoder.layers:
                x = self._layer(x, lyr)
            outs.append(ops.linear(x, self.output_proj.weight, self.output_proj.bias.float()))
        return torch.stack(outs, dim=0)
