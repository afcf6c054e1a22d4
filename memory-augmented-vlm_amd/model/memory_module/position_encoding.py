"""Temporal positional encoding on the HIP path.

Drop-in for ``TemporalPositionalEncoding`` (llava/model/memory_module/position_encoding.py:13-80): same
constructor, same ``frame_embed`` buffer / embedding (state-dict key ``positional_encoding.frame_embed``), same
``forward(x, frame_indices=None)`` and the same ``ValueError`` conditions.  The gather + broadcast add runs as
one vectorised HIP kernel (mavlm_pe_add) instead of index_select + add."""
import math

import torch
import torch.nn as nn

from ... import _capi as capi
from ... import _ops as ops


class TemporalPositionalEncoding(nn.Module):
    def __init__(self, max_frames, embed_dim, learnable=True):
        super().__init__()
        self.max_frames = max_frames
        self.embed_dim = embed_dim
        self.learnable = learnable
        if learnable:
            self.frame_embed = nn.Embedding(max_frames, embed_dim)
        else:
            # pe[p,2i] = sin(p * 10000^(-2i/D)), pe[p,2i+1] = cos(same), float32 (position_encoding.py:29-35)
            pos = torch.arange(0, max_frames).unsqueeze(1).float()
            freq = torch.exp(torch.arange(0, embed_dim, 2).float() * -(math.log(10000.0) / embed_dim))
            table = torch.zeros(max_frames, embed_dim, dtype=torch.float32)
            table[:, 0::2] = torch.sin(pos * freq)
            table[:, 1::2] = torch.cos(pos * freq)
            self.register_buffer("frame_embed", table)

    def table(self) -> torch.Tensor:
        return self.frame_embed.weight if self.learnable else self.frame_embed

    def check_indices(self, indices_cpu: torch.Tensor):
        """Range check with the reference's messages (position_encoding.py:73-76), on a host copy."""
        if indices_cpu.numel() == 0:
            return
        hi, lo = int(indices_cpu.max()), int(indices_cpu.min())
        if hi >= self.max_frames:
            raise ValueError(f"indices exceed max_frames: max {hi} vs limit {self.max_frames}")
        if lo < 0:
            raise ValueError(f"indices contains negative values: min {lo}")

    def forward(self, x, frame_indices=None, indices_checked=False):
        if x.dim() not in (3, 4):
            raise ValueError(f"Expected 3D or 4D input, got {x.dim()}D.")
        lead = x.shape[:-2]                                   # (T,) or (B,T)
        if frame_indices is None:
            t = torch.arange(lead[-1])
            frame_indices = t if x.dim() == 3 else t.expand(*lead)
            indices_checked = lead[-1] <= self.max_frames
            if not indices_checked:
                self.check_indices(frame_indices)
        elif not indices_checked:
            self.check_indices(frame_indices.detach().cpu())  # one host sync if the indices live on the GPU
        if not x.is_cuda:
            raise capi.MavlmError("TemporalPositionalEncoding: input is not on a GPU (no CPU fallback)")
        idx = frame_indices.reshape(-1).to(device=x.device, dtype=torch.int64)
        tab = self.table().to(device=x.device, dtype=x.dtype)  # `.to(x.dtype)`, position_encoding.py:58
        x3 = x.contiguous().reshape(-1, x.shape[-2], x.shape[-1])
        if torch.is_grad_enabled() and (x.requires_grad or (self.learnable and tab.requires_grad)):
            # a gradient has to flow (learnable table, or frame features that are not detached): the gather + add is
            # index plumbing, left to autograd; the reference configuration (fixed table, detached features,
            # llava_arch.py:150,302) never takes this branch
            return (x3 + tab[idx][:, None, :]).reshape(x.shape)
        out = ops.row_add(x3, tab.contiguous(), idx=idx)
        return out.reshape(x.shape)
