"""Inactive variant: `TemporalGRUEncoder` of llava/model/memory_module/bigru.py:14-75 (its
instantiation is commented out at llava_arch.py:151).  Same constructor, same parameters (`gru.weight_ih_l0`, ...:
a real nn.GRU holds them) and the optional `temporal_pe` buffer; forward on the HIP kernels:

  per-frame mean over the patches (frame_mean_kernel) -> [+ sinusoidal PE] -> the input half of both directions as ONE
  MFMA GEMM (W_ih x + b_ih, fp32 out) -> the recurrent half (gru_seq_kernel, one workgroup per direction, state in
  LDS, fp32 gate math) -> broadcast residual add over the patches (row_add_kernel).
Forward only (eval); one layer, hidden_size <= 512 and 3*hidden*directions a multiple of 128 (896/448, 1024/512).
"""
import math

import torch
import torch.nn as nn

from ... import _capi as capi
from ... import _ops as ops


def build_sine_time_table(max_frames: int, dim: int, device=None):
    """pe[t,2i] = sin(t * 10000^(-2i/dim)), pe[t,2i+1] = cos(same)  (bigru.py:5-12)."""
    pos = torch.arange(max_frames, dtype=torch.float32, device=device).unsqueeze(1)
    freq = torch.exp(torch.arange(0, dim, 2, device=device) * (-math.log(10000.0) / dim))
    pe = torch.zeros(max_frames, dim, device=device)
    pe[:, 0::2] = torch.sin(pos * freq)
    pe[:, 1::2] = torch.cos(pos * freq)
    return pe


class TemporalGRUEncoder(nn.Module):
    def __init__(self, input_dim: int = 896, hidden_size: int = 448, num_layers: int = 1, bidirectional: bool = True,
                 max_frames: int = 300, use_positional_encoding: bool = False):
        super().__init__()
        self.use_pe = use_positional_encoding
        self.gru = nn.GRU(input_size=input_dim, hidden_size=hidden_size, num_layers=num_layers,
                          bidirectional=bidirectional, batch_first=False)
        ndir = 2 if bidirectional else 1
        if num_layers != 1 or hidden_size > 512 or hidden_size % 8 or (3 * hidden_size * ndir) % 128 or \
                hidden_size * ndir != input_dim or input_dim % 64:
            raise capi.MavlmError("TemporalGRUEncoder (HIP): one layer, hidden_size <= 512, 3*hidden*directions a "
                                  "multiple of 128 and hidden*directions == input_dim (the residual add needs it)")
        self.ndir, self.hidden = ndir, hidden_size
        if self.use_pe:
            self.register_buffer("temporal_pe", build_sine_time_table(max_frames, input_dim, device=torch.device("cpu")))

    def _packed(self, dtype):
        sfx = ["", "_reverse"][:self.ndir]
        g = self.gru
        w_ih = torch.cat([getattr(g, "weight_ih_l0" + s) for s in sfx], dim=0).to(dtype).contiguous()
        b_ih = torch.cat([getattr(g, "bias_ih_l0" + s) for s in sfx], dim=0).float().contiguous()
        w_hh = torch.stack([getattr(g, "weight_hh_l0" + s) for s in sfx], dim=0).to(dtype).contiguous()
        b_hh = torch.stack([getattr(g, "bias_hh_l0" + s) for s in sfx], dim=0).float().contiguous()
        return w_ih, b_ih, w_hh, b_hh

    @torch.no_grad()
    def forward(self, visual_feats: torch.Tensor) -> torch.Tensor:
        if visual_feats.dim() != 3:
            raise capi.MavlmError("TemporalGRUEncoder expects [F, P, D]")
        if not visual_feats.is_cuda:
            raise capi.MavlmError("TemporalGRUEncoder: input is not on a GPU; the HIP path has no CPU fallback")
        F = visual_feats.shape[0]
        x = visual_feats.contiguous()
        vecs = ops.frame_mean(x)                                                           # bigru.py:50
        if self.use_pe:                                                                    # :53-55
            if F > self.temporal_pe.shape[0]:
                raise capi.MavlmError("more frames than max_frames of the temporal table")
            vecs = ops.row_add(vecs[:, None, :], self.temporal_pe.to(device=x.device, dtype=x.dtype).contiguous(),
                               idx=torch.arange(F, device=x.device))[:, 0, :]
        w_ih, b_ih, w_hh, b_hh = self._packed(x.dtype)
        xg = ops.linear(vecs.contiguous(), w_ih, b_ih, capi.EPI_F32)                       # input half of :68
        ctx = ops.gru_sequence(xg, w_hh, b_hh, self.hidden, self.ndir)                     # recurrent half
        return ops.row_add(x, ctx, idx=torch.arange(F, device=x.device))                   # :71-74
