"""Step before the path: the vision projector that feeds `get_2dPool`.

Mirrors `llava/model/multimodal_projector/builder.py:32-65` for the projector types the memory path is trained with
(`mlp{N}x_gelu`, `linear`, `identity`).  State-dict keys are those of the reference's modules (`nn.Sequential`:
`0.weight, 0.bias, 2.weight, ...`; `linear`: a bare `nn.Linear`, `weight, bias`), so a LLaVA-OneVision `mm_projector.*` checkpoint loads unchanged; forward runs on
the MFMA GEMMs of the HIP library with the exact-erf GELU fused into the producing epilogue.  `pooler` and the
`res` variants are not used by the memory-path configurations and raise.
"""
import re

import torch.nn as nn

from ... import _capi as capi
from ... import _ops as ops


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


class MlpGeluProjector(nn.Sequential):
    """Linear(mm_hidden, D) [GELU Linear(D, D)]*(depth-1).  The reference detaches this module's output
    (llava_arch.py:302), so the hot path only needs the forward; with autograd recording and trainable parameters the
    same GEMMs run as autograd Functions (backward in HIP, _autograd.py) so that nothing is silently constant."""

    def __init__(self, mm_hidden: int, hidden: int, depth: int):
        mods = [nn.Linear(mm_hidden, hidden)]
        for _ in range(1, depth):
            mods += [nn.GELU(), nn.Linear(hidden, hidden)]
        super().__init__(*mods)

    def forward(self, x):
        lin = [m for m in self if isinstance(m, nn.Linear)]
        y = x.reshape(-1, x.shape[-1])
        if not y.is_contiguous():
            y = y.contiguous()
        import torch
        train = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if train:
            from ... import _autograd as ag
        for i, m in enumerate(lin):
            last = i == len(lin) - 1
            if train:
                y = ag.LinearFn.apply(y, m.weight.to(y.dtype), m.bias, ag.ACT_NONE if last else ag.ACT_GELU)
            else:
                y = ops.linear(y, m.weight, m.bias.float(), capi.EPI_BIAS if last else capi.EPI_GELU)
        return y.reshape(*x.shape[:-1], y.shape[-1])


class LinearProjector(nn.Linear):
    """`mm_projector_type="linear"`: the reference returns a bare `nn.Linear` (builder.py:35-36), state-dict keys
    `weight` / `bias` - kept, so such a checkpoint loads; forward = one HIP GEMM."""

    def forward(self, x):
        import torch
        y = x.reshape(-1, x.shape[-1])
        if not y.is_contiguous():
            y = y.contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ... import _autograd as ag
            y = ag.LinearFn.apply(y, self.weight.to(y.dtype), self.bias, ag.ACT_NONE)
        else:
            y = ops.linear(y, self.weight, self.bias.float(), capi.EPI_BIAS)
        return y.reshape(*x.shape[:-1], y.shape[-1])


def build_vision_projector(config, delay_load=False, **kwargs):
    kind = getattr(config, "mm_projector_type", "linear")
    if kind == "linear":
        return LinearProjector(config.mm_hidden_size, config.hidden_size)
    m = re.match(r"^mlp(\d+)x_gelu$", kind)
    if m:
        return MlpGeluProjector(config.mm_hidden_size, config.hidden_size, int(m.group(1)))
    if kind == "identity":
        return IdentityMap()
    if kind == "pooler" or re.match(r"^mlp(\d+)x_res(\d+)x_gelu$", kind):
        raise NotImplementedError(f"mm_projector_type {kind!r}: not used by the memory-path configurations")
    raise ValueError(f"Unknown projector type: {kind}")
