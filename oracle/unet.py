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

# bf16-EMULATING mode (VERDICT r4 #3b; tests/test_gpu_round5.py): with ROUND_OPERANDS = "bf16" every matrix product that bf16 mode (said_set_precision) runs on
# bf16 operands takes BOTH operands rounded to bf16 (round-to-nearest-even, as the kernels' conversions) and accumulates in float64; everything the kernels keep in
# fp32 stays fp32 here: statistics, normalisations, activations, softmax, biases, residual sums, conv_in, the time embedding, the cross-attention K / V projection and
# its banded products.  The folded (proj_out o ff.net.2) product is formed as the engine forms it (in double, then rounded once: engine.cpp make_pw "__ffproj").
# What it cannot reproduce: the summation order, and roundings that depend on a kernel's tiling (the online softmax rounds UNNORMALISED probabilities per key
# slice; large batches additionally store activations in bf16 between kernels: ROUND_STORES).  Agreement with the HIP bf16 path is therefore statistical, but at the
# level of single rounding flips instead of the rounding itself.
ROUND_OPERANDS = None
ROUND_STORES = False   # large-batch bf16 schedule: the hidden state between kernels is stored in bf16 (rgemm.hip / stchain_kernel<bf16>)


def _r(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float64)


def _st(x: torch.Tensor) -> torch.Tensor:
    """A tensor one kernel stores and the next one reads (large-batch bf16 schedule: bf16 storage)."""
    return x.to(torch.bfloat16).to(torch.float32) if (ROUND_OPERANDS == "bf16" and ROUND_STORES) else x


def _linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    if ROUND_OPERANDS != "bf16":
        return F.linear(x, w, b)
    y = F.linear(_r(x), _r(w)).float()
    return y if b is None else y + b


def _conv1d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, padding: int = 0) -> torch.Tensor:
    if ROUND_OPERANDS != "bf16":
        return F.conv1d(x, w, b, padding=padding)
    y = F.conv1d(_r(x), _r(w), None, padding=padding).float()
    return y if b is None else y + b[None, :, None]


def _mm(eq: str, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if ROUND_OPERANDS != "bf16":
        return torch.einsum(eq, a, b)
    return torch.einsum(eq, _r(a), _r(b)).float()


ATTN_KS = 4          # key slices of the self-attention kernel being emulated (attn_kernel<.., KS, ..>: 4 at small batch; None: plain softmax on rounded operands)


def attn_bf16_online(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, KS: int) -> torch.Tensor:
    """attn_kernel<PM = 1> restated (attn.hip): S = bf16(k) . bf16(q) in raw units, online softmax over 32-key tiles — wave w of KS takes tiles w, w + KS, ... —
    with the running maximum in raw units and the scale folded into the exponent, the row sums from the UNROUNDED probabilities, P rounded to bf16 for P . bf16(v),
    fixed-order merge of the waves' partial results.  q, k, v: (B * heads, T, 32) fp32."""
    BH, T, D = q.shape
    c2 = torch.tensor((D ** -0.5) * 1.4426950408889634, dtype=torch.float32)
    r = lambda t: t.to(torch.bfloat16).to(torch.float64)
    s_raw = torch.einsum("bid,bjd->bij", r(q), r(k)).float()
    vb = r(v)
    nkt = (T + 31) // 32
    ms, ls, os_ = [], [], []
    for w in range(KS):
        m = torch.full((BH, T), -1.0e30)
        lsum = torch.zeros(BH, T)
        o = torch.zeros(BH, T, D)
        for kt in range(w, nkt, KS):
            j0, j1 = kt * 32, min(kt * 32 + 32, T)
            st = s_raw[:, :, j0:j1]
            mn = torch.maximum(m, st.max(dim=-1).values)
            alpha = torch.exp2((m - mn) * c2)
            pr = torch.exp2(st * c2 + (-mn * c2)[..., None])
            lsum = lsum * alpha + pr.sum(dim=-1)
            o = o * alpha[..., None] + torch.einsum("bij,bjd->bid", r(pr), vb[:, j0:j1]).float()
            m = mn
        ms.append(m); ls.append(lsum); os_.append(o)
    M = torch.stack(ms).max(dim=0).values
    L = torch.zeros(BH, T)
    out = torch.zeros(BH, T, D)
    for w in range(KS):
        f = torch.exp2((ms[w] - M) * c2)
        L = L + ls[w] * f
        out = out + os_[w] * f[..., None]
    return out * (1.0 / L)[..., None]


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
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    if ROUND_OPERANDS == "bf16":   # (bias and embedding row join the fp32 accumulator in the kernel's epilogue)
        h = _conv1d(h, sd[p + ".in_layers.2.weight"], None, padding=1) + (sd[p + ".in_layers.2.bias"][None] + e)[..., None]
    else:
        h = F.conv1d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
        h = h + e[..., None]
    h = _st(h)
    _t(p + ":mid", h)
    h = F.group_norm(h.float(), 32, sd[p + ".out_layers.0.weight"], sd[p + ".out_layers.0.bias"], eps=1e-5)
    h = F.silu(h)
    h = _conv1d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = _conv1d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    _t(p + ":out", x + h)
    return _st(x + h)


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
    q = _linear(x, sd[p + ".to_q.weight"])
    ctx = x if context is None else context
    if context is None:
        k = _linear(ctx, sd[p + ".to_k.weight"])
        v = _linear(ctx, sd[p + ".to_v.weight"])
    else:   # the context's K / V projection is an fp32 GEMM in either mode (run_kv)
        k = F.linear(ctx, sd[p + ".to_k.weight"])
        v = F.linear(ctx, sd[p + ".to_v.weight"])
    b, n, _ = q.shape
    self_attn = context is None
    if self_attn:
        q, k, v = _st(q), _st(k), _st(v)
    elif ROUND_OPERANDS == "bf16" and ROUND_STORES:   # stchain_kernel<bf16> keeps the window tiles of K / V in bf16 (q stays fp32)
        k, v = _st(k), _st(v)

    def split(t: torch.Tensor) -> torch.Tensor:  # b n (h d) -> (b h) n d
        return t.reshape(b, t.shape[1], h, -1).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], -1)

    if context is None:
        _t(p + ":q", q); _t(p + ":k", k); _t(p + ":v", v)
    q, k, v = split(q), split(k), split(v)
    if self_attn and ROUND_OPERANDS == "bf16" and ATTN_KS:
        out = attn_bf16_online(q, k, v, ATTN_KS)
        out = _st(out.reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1))
        _t(p + ":attn", out)
        return _linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    sim = (_mm("bid,bjd->bij", q, k) if self_attn else torch.einsum("bid,bjd->bij", q, k)) * (HEAD_DIM ** -0.5)
    if mask is not None:
        m = mask[:, None].expand(b, h, *mask.shape[1:]).reshape(b * h, *mask.shape[1:])
        sim = sim.masked_fill(m, -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    out = _mm("bij,bjd->bid", attn, v) if self_attn else torch.einsum("bij,bjd->bid", attn, v)
    out = out.reshape(b, h, n, -1).permute(0, 2, 1, 3).reshape(b, n, -1)
    if self_attn:
        out = _st(out)
    _t(p + ":attn", out)
    return _linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ldm/attention.py:25-51 — GEGLU with exact (erf) GELU."""
    y = _linear(x, sd[p + ".net.0.proj.weight"], sd[p + ".net.0.proj.bias"])
    a, gate = y.chunk(2, dim=-1)
    y = a * F.gelu(gate)
    _t(p + ":geglu", y)
    if ROUND_OPERANDS == "bf16":
        return y   # (the caller multiplies [y ; x2] with the folded weights)
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
    y = feed_forward(sd, p + ".ff", F.layer_norm(x, (C,), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"]))
    if ROUND_OPERANDS == "bf16":
        return y, x   # GEGLU product and x2: operands of the folded proj_out
    x = y + x
    _t(p + ":x3", x)
    return x


def spatial_transformer(sd: SD, p: str, x: torch.Tensor, context: Optional[torch.Tensor]) -> torch.Tensor:
    """ldm/attention.py:223-234 — GroupNorm eps 1e-6, no proj_in, 1x1 proj_out."""
    x_in = x
    h = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-6)
    h = h.transpose(1, 2)
    h = transformer_block(sd, p + ".transformer_blocks.0", h, context)
    if ROUND_OPERANDS == "bf16":
        # proj(F2 y + b2 + x2) + bp = (P F2) y + P x2 + (P b2 + bp): the products P F2 and P b2 + bp in double, each rounded ONCE (engine.cpp: "__ffproj")
        y, x2 = h
        b0 = p + ".transformer_blocks.0.ff.net.2."
        P = sd[p + ".proj_out.weight"].reshape(MODEL_CH, MODEL_CH).double()
        PF = (P @ sd[b0 + "weight"].double()).float()
        PB = (P @ sd[b0 + "bias"].double() + sd[p + ".proj_out.bias"].double()).float()
        out = _linear(y, PF) + _linear(x2, P.float()) + PB
        out = out.transpose(1, 2)
        _t(p + ":out", out + x_in)
        return _st(out + x_in)
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
    h = _st(h)
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
    return _conv1d(h, sd["model.out.2.weight"], sd["model.out.2.bias"], padding=1)


def unet1d_forward(sd: SD, sample: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
    """unet_1d_condition.py:51-77 — (B,T,C) in/out."""
    out = unet_model_forward(sd, sample.transpose(1, 2), timestep, encoder_hidden_states)
    return out.transpose(1, 2)
