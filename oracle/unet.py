"""CPU restatement of the conditional UNet1D denoiser — TEST INFRASTRUCTURE.

Functional PyTorch-CPU fp32 code driven by a reference-keyed state dict
(``model.*`` names).  Each function cites the reference lines it
follows under /root/reference/said/model/.  Pinned by tests/golden (G1-G4).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

MODEL_CH = 192
HEADS = 6
HEAD_DIM = 32

# Optional stage trace for kernel bring-up (tests/debug_stages.py): when set to a list, every
# tensor that one HIP kernel launch materialises is appended as (name, tensor), in launch order.
TRACE = None


def _t(name: str, x: torch.Tensor) -> None:
    if TRACE is not None:
        TRACE.append((name, x.detach().clone()))


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """ldm/util.py:66-90 — ``[cos(t·f), sin(t·f)]``, cos half first."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def time_embed(sd: SD, timesteps: torch.Tensor) -> torch.Tensor:
    """ldm/openaimodel.py:464-468, 690-691."""
    e = timestep_embedding(timesteps, MODEL_CH)
    e = F.linear(e, sd["model.time_embed.0.weight"], sd["model.time_embed.0.bias"])
    e = F.silu(e)
    return F.linear(e, sd["model.time_embed.2.weight"], sd["model.time_embed.2.bias"])


def res_block(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """ldm/openaimodel.py:205-227 (no up/down, use_scale_shift_norm=False)."""
    h = F.group_norm(x.float(), 32, sd[p + ".in_layers.0.weight"], sd[p + ".in_layers.0.bias"], eps=1e-5)
    h = F.silu(h)
    h = F.conv1d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    h = h + e[..., None]
    _t(p + ":mid", h)
    h = F.group_norm(h.float(), 32, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], eps=1e-5)
    h = F.silu(h)
    h = F.conv1d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = F.conv1d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    _t(p + ":out", x + h)
    return x + h


def alignment_mask(batch: int, x_len: int, c_len: int, pad: int = 1) -> torch.Tensor:
    """ldm/attention.py:170-189 — True = masked; Python banker's ``round``."""
    ratio = c_len / x_len
    kh = ratio / 2 + pad
    m = torch.ones(batch, x_len, c_len, dtype=torch.bool)
    for i in range(x_len):
        mid = (i + 0.5) * ratio
        lo = max(round(mid - kh), 0)
        hi = min(round(mid + kh), c_len)
        m[:, i, lo:hi] = False
    return m


def cross_attention(sd: SD, p: str, x: torch.Tensor, context: Optional[torch.Tensor],
                    mask: Optional[torch.Tensor]) -> torch.Tensor:
    """ldm/attention.py:86-128 — scale applied after QKᵀ; masked_fill(-finfo.max)."""
    h = HEADS
    q = F.linear(x, sd[p + ".to_q.weight"])
    ctx = x if context is None else context
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    b, n, _ = q.shape

    def split(t: torch.Tensor) -> torch.Tensor:  # b n (h d) -> (b h) n d
        return t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)

    if context is None:
        _t(p + ":q", q); _t(p + ":k", k); _t(p + ":v", v)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (HEAD_DIM ** -0.5)
    if mask is not None:
        m = mask[:, None].expand(b, h, *mask.shape[1:]).reshape(b * h, *mask.shape[1:])
        sim = sim.masked_fill(m, -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
    _t(p + ":attn", out)
    return F.linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ldm/attention.py:25-51 — GEGLU with exact (erf) GELU."""
    y = F.linear(x, sd[p + ".net.0.proj.weight"], sd[p + ".net.0.proj.bias"])
    a, gate = y.chunk(2, dim=-1)
    y = a * F.gelu(gate)
    _t(p + ":geglu", y)
    return F.linear(y, sd[p + ".net.2.weight"], sd[p + ".net.2.bias"])


def transformer_block(sd: SD, p: str, x: torch.Tensor, context: Optional[torch.Tensor]) -> torch.Tensor:
    """ldm/attention.py:167-193."""
    C = x.shape[-1]
    x = cross_attention(sd, p + ".attn1", F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]),
                        None, None) + x
    _t(p + ":x1", x)
    mask = None
    if context is not None:
        mask = alignment_mask(x.shape[0], x.shape[1], context.shape[1])
    x = cross_attention(sd, p + ".attn2", F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]),
                        context, mask) + x
    _t(p + ":x2", x)
    x = feed_forward(sd, p + ".ff", F.layer_norm(x, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])) + x
    _t(p + ":x3", x)
    return x


def spatial_transformer(sd: SD, p: str, x: torch.Tensor, context: Optional[torch.Tensor]) -> torch.Tensor:
    """ldm/attention.py:223-234 — GroupNorm eps 1e-6, no proj_in, 1x1 proj_out."""
    x_in = x
    h = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-6)
    h = h.transpose(1, 2)
    h = transformer_block(sd, p + ".transformer_blocks.0", h, context)
    h = h.transpose(1, 2)
    h = F.conv1d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    _t(p + ":out", h + x_in)
    return h + x_in


def unet_model_forward(sd: SD, x: torch.Tensor, timesteps: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
    """ldm/openaimodel.py:677-709 with the block wiring of unet_1d_condition.py:36-49.
    ``x`` is (B, C, T)."""
    emb = time_embed(sd, timesteps)
    hs = []
    h = F.conv1d(x.float(), sd["model.input_blocks.0.0.weight"], sd["model.input_blocks.0.0.bias"], padding=1)
    _t("conv_in", h)
    hs.append(h)
    h = res_block(sd, "model.input_blocks.1.0", h, emb)
    h = spatial_transformer(sd, "model.input_blocks.1.1", h, context)
    hs.append(h)
    h = res_block(sd, "model.middle_block.0", h, emb)
    h = spatial_transformer(sd, "model.middle_block.1", h, context)
    h = res_block(sd, "model.middle_block.2", h, emb)
    for i in range(2):
        h = torch.cat([h, hs.pop()], dim=1)
        h = res_block(sd, f"model.output_blocks.{i}.0", h, emb)
        h = spatial_transformer(sd, f"model.output_blocks.{i}.1", h, context)
    h = F.group_norm(h.float(), 32, sd["model.out.0.weight"], sd["model.out.0.bias"], eps=1e-5)
    h = F.silu(h)
    return F.conv1d(h, sd["model.out.2.weight"], sd["model.out.2.bias"], padding=1)


def unet1d_forward(sd: SD, sample: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
    """unet_1d_condition.py:51-77 — (B,T,C) in/out."""
    out = unet_model_forward(sd, sample.transpose(1, 2), timestep, encoder_hidden_states)
    return out.transpose(1, 2)
