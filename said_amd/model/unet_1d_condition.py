"""Conditional 1D UNet denoiser — parameter container + HIP-engine forward.

Drop-in for the reference's ``said.model.unet_1d_condition.UNet1DConditionModel``
(/root/reference/said/model/unet_1d_condition.py:8-77) and the vendored LDM UNet
it instantiates (said/model/ldm/openaimodel.py:397-709): identical constructor,
``in_channels`` / ``out_channels`` / ``cross_attention_dim`` attributes, identical
``state_dict()`` keys (160 tensors under ``model.*``) and initialisation
(``zero_module`` layers start at zero).  The module tree below only *holds*
parameters; ``forward`` hands them to the gfx950 engine (``said_unet_forward``).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import _engine

MODEL_CHANNELS = 192
NUM_HEAD_CHANNELS = 32


def _zero(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


class _Res(nn.Module):
    """Parameter layout of the reference ResBlock (openaimodel.py:132-194)."""

    def __init__(self, channels: int, emb_channels: int, dropout: float, out_channels: int):
        super().__init__()
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Conv1d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        _zero(nn.Conv1d(out_channels, out_channels, 3, padding=1)))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv1d(channels, out_channels, 1)


class _Attn(nn.Module):
    """Parameter layout of CrossAttention (ldm/attention.py:69-84)."""

    def __init__(self, query_dim: int, context_dim: Optional[int], heads: int, dim_head: int, dropout: float):
        super().__init__()
        inner = heads * dim_head
        context_dim = query_dim if context_dim is None else context_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class _GEGLUProj(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class _FF(nn.Module):
    def __init__(self, dim: int, dropout: float):
        super().__init__()
        self.net = nn.Sequential(_GEGLUProj(dim, dim * 4), nn.Dropout(dropout), nn.Linear(dim * 4, dim))


class _TBlock(nn.Module):
    """Parameter layout of BasicTransformerBlock (ldm/attention.py:131-163)."""

    def __init__(self, dim: int, heads: int, d_head: int, dropout: float, context_dim: int):
        super().__init__()
        self.attn1 = _Attn(dim, None, heads, d_head, dropout)
        self.ff = _FF(dim, dropout)
        self.attn2 = _Attn(dim, context_dim, heads, d_head, dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)


class _ST(nn.Module):
    """Parameter layout of SpatialTransformer (ldm/attention.py:204-221): no proj_in."""

    def __init__(self, channels: int, heads: int, d_head: int, context_dim: int):
        super().__init__()
        self.norm = nn.GroupNorm(32, channels, eps=1e-6, affine=True)
        self.transformer_blocks = nn.ModuleList([_TBlock(heads * d_head, heads, d_head, 0.0, context_dim)])
        self.proj_out = _zero(nn.Conv1d(channels, channels, 1))


class _UNetParams(nn.Module):
    """``UNetModel(dims=1, model_channels=192, num_res_blocks=1, attention_resolutions=(1,),
    channel_mult=(1,), num_head_channels=32, use_spatial_transformer=True, transformer_depth=1)``
    (unet_1d_condition.py:36-49)."""

    def __init__(self, in_channels: int, out_channels: int, context_dim: int, dropout: float):
        super().__init__()
        C, E = MODEL_CHANNELS, 4 * MODEL_CHANNELS
        heads = C // NUM_HEAD_CHANNELS
        self.time_embed = nn.Sequential(nn.Linear(C, E), nn.SiLU(), nn.Linear(E, E))
        self.input_blocks = nn.ModuleList([
            nn.Sequential(nn.Conv1d(in_channels, C, 3, padding=1)),
            nn.Sequential(_Res(C, E, dropout, C), _ST(C, heads, NUM_HEAD_CHANNELS, context_dim)),
        ])
        self.middle_block = nn.Sequential(_Res(C, E, dropout, C), _ST(C, heads, NUM_HEAD_CHANNELS, context_dim), _Res(C, E, dropout, C))
        self.output_blocks = nn.ModuleList([
            nn.Sequential(_Res(2 * C, E, dropout, C), _ST(C, heads, NUM_HEAD_CHANNELS, context_dim)) for _ in range(2)
        ])
        self.out = nn.Sequential(nn.GroupNorm(32, C), nn.SiLU(), _zero(nn.Conv1d(C, out_channels, 3, padding=1)))


class UNet1DConditionModel(nn.Module):
    """Conditional 1D UNet model"""

    def __init__(self, in_channels: int, out_channels: int, cross_attention_dim: int, dropout: float = 0.1) -> None:
        super().__init__()
        if in_channels != out_channels:
            raise ValueError("the SAiD denoiser uses in_channels == out_channels")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.cross_attention_dim = cross_attention_dim
        self.model = _UNetParams(in_channels, out_channels, cross_attention_dim, dropout)
        self._owner = None          # set by SAID: the pipeline's engine is shared
        self._own_engine = None     # standalone use
        self._own_key = None

    def _engine_for(self, batch_eff: int, frames: int) -> _engine.Engine:
        if self._owner is not None:
            return self._owner()._get_engine(batch_eff, frames)
        p = next(self.parameters())
        key = (str(p.device), sum(q._version for q in self.parameters()))
        e = self._own_engine
        if e is None or self._own_key != key or e.max_batch_eff < batch_eff or e.max_frames < frames:
            if e is not None:
                e.close()
            e = _engine.Engine(p.device, max(batch_eff, 2), max(frames, 64), self.in_channels, self.cross_attention_dim)
            sd = {"denoiser." + k: v for k, v in self.state_dict().items()}
            sd["null_cond_emb"] = torch.zeros(1, 1, self.cross_attention_dim)
            e.load_weights(sd)
            self._own_engine, self._own_key = e, key
        return e

    def forward(self, sample: torch.FloatTensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor) -> torch.FloatTensor:
        """(B, T, C) noisy coefficients, (B,)/(1,)/() timesteps, (B, S, D) audio tokens → (B, T, C)."""
        B, T, _ = sample.shape
        ts = torch.as_tensor(timestep).reshape(-1)
        if ts.numel() == 1 and B > 1:
            ts = ts.repeat(B)
        assert ts.shape[0] == B, "one timestep per sample"
        return self._engine_for(B, max(T, encoder_hidden_states.shape[1])).unet_forward(sample, ts, encoder_hidden_states)
