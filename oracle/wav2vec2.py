"""CPU restatement of the audio encoder — TEST INFRASTRUCTURE.

``ModifiedWav2Vec2Model.forward`` (/root/reference/said/model/wav2vec2.py:13-82)
over the **third-party** ``transformers==4.30.2`` Wav2Vec2 base architecture
(pinned at reference pyproject.toml:29; absent from /root/reference).  The
restatement follows that library's published eval-mode algorithm for the default
``Wav2Vec2Config()``: group-norm feature extractor, post-LN encoder, weight-normed
grouped positional conv, q scaled by head_dim**-0.5 before QKᵀ.  It is pinned
(tests/golden G5) against the reference's subclass run on the transformers
version installed in the build container (5.15.0), whose layer numerics are the
same stock torch ops.  State-dict keys use the 4.30.2 naming
(``…pos_conv_embed.conv.weight_g/weight_v``).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
HEADS = 12
LN_EPS = 1e-5


def feature_extractor(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """7 conv layers, no bias; layer 0 followed by GroupNorm(512 groups) (per-channel
    over time); exact GELU after each.  (B,Ta) -> (B,512,L)."""
    h = x[:, None]
    for i, s in enumerate(CONV_STRIDE):
        h = F.conv1d(h, sd[f"feature_extractor.conv_layers.{i}.conv.weight"], None, stride=s)
        if i == 0:
            c = h.shape[1]
            h = F.group_norm(h, c, sd["feature_extractor.conv_layers.0.layer_norm.weight"],
                             sd["feature_extractor.conv_layers.0.layer_norm.bias"], eps=1e-5)
        h = F.gelu(h)
    return h


def pos_conv_weight(sd: SD) -> torch.Tensor:
    """weight_norm(dim=2): w = g · v / ‖v‖ with the norm over dims (0,1)."""
    g = sd["encoder.pos_conv_embed.conv.weight_g"]
    v = sd["encoder.pos_conv_embed.conv.weight_v"]
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def attention(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    b, t, c = x.shape
    d = c // HEADS
    q = F.linear(x, sd[p + ".q_proj.weight"], sd[p + ".q_proj.bias"]) * (d ** -0.5)
    k = F.linear(x, sd[p + ".k_proj.weight"], sd[p + ".k_proj.bias"])
    v = F.linear(x, sd[p + ".v_proj.weight"], sd[p + ".v_proj.bias"])

    def split(z: torch.Tensor) -> torch.Tensor:
        return z.view(b, t, HEADS, d).transpose(1, 2)

    w = torch.matmul(split(q), split(k).transpose(2, 3)).softmax(dim=-1)
    o = torch.matmul(w, split(v)).transpose(1, 2).reshape(b, t, c)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def encoder_layer(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Post-LN layer (do_stable_layer_norm=False)."""
    c = x.shape[-1]
    x = x + attention(sd, p + ".attention", x)
    x = F.layer_norm(x, (c,), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], LN_EPS)
    f = F.linear(x, sd[p + ".feed_forward.intermediate_dense.weight"], sd[p + ".feed_forward.intermediate_dense.bias"])
    f = F.gelu(f)
    f = F.linear(f, sd[p + ".feed_forward.output_dense.weight"], sd[p + ".feed_forward.output_dense.bias"])
    x = x + f
    return F.layer_norm(x, (c,), sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"], LN_EPS)


def num_layers(sd: SD) -> int:
    n = 0
    while f"encoder.layers.{n}.layer_norm.weight" in sd:
        n += 1
    return n


def wav2vec2_forward(sd: SD, input_values: torch.Tensor, num_frames: Optional[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (last_hidden_state (B,F,768), conv features before interpolation (B,512,L))."""
    feats = feature_extractor(sd, input_values.float())
    h = feats
    if num_frames is not None:  # wav2vec2.py:41-44
        h = F.interpolate(h, size=num_frames, align_corners=True, mode="linear")
    h = h.transpose(1, 2)
    c = h.shape[-1]
    h = F.layer_norm(h, (c,), sd["feature_projection.layer_norm.weight"], sd["feature_projection.layer_norm.bias"], LN_EPS)
    h = F.linear(h, sd["feature_projection.projection.weight"], sd["feature_projection.projection.bias"])
    # encoder
    w = pos_conv_weight(sd)
    k = w.shape[-1]
    groups = h.shape[-1] // w.shape[1]
    pos = F.conv1d(h.transpose(1, 2), w, sd["encoder.pos_conv_embed.conv.bias"], padding=k // 2, groups=groups)
    if k % 2 == 0:
        pos = pos[:, :, :-1]
    pos = F.gelu(pos).transpose(1, 2)
    h = h + pos
    H = h.shape[-1]
    h = F.layer_norm(h, (H,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], LN_EPS)
    for l in range(num_layers(sd)):
        h = encoder_layer(sd, f"encoder.layers.{l}", h)
    return h, feats
