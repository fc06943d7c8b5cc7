"""Audio encoder — parameter container + HIP-engine forward.

Drop-in for ``said.model.wav2vec2.ModifiedWav2Vec2Model``
(/root/reference/said/model/wav2vec2.py:13-82), i.e. HF ``Wav2Vec2Model`` (base
architecture, ``transformers==4.30.2`` naming) whose conv features are linearly
interpolated to ``num_frames`` before the projection.  ``state_dict()`` matches the
reference checkpoint key-for-key (211 tensors for the default config, including
``masked_spec_embed`` and ``encoder.pos_conv_embed.conv.weight_g/weight_v``); the
newer ``…parametrizations.weight.original0/1`` spelling is accepted on load.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn

from .. import _engine


@dataclass
class AudioConfig:
    """The subset of ``transformers.Wav2Vec2Config()`` defaults this path depends on."""
    hidden_size: int = 768
    output_hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    feat_extract_norm: str = "group"
    conv_bias: bool = False
    do_stable_layer_norm: bool = False


@dataclass
class Wav2Vec2BaseModelOutput:
    last_hidden_state: torch.Tensor
    extract_features: Optional[torch.Tensor] = None
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None


class _ConvLayer(nn.Module):
    def __init__(self, cin: int, cout: int, k: int, s: int, group_norm: bool):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, stride=s, bias=False)
        if group_norm:
            self.layer_norm = nn.GroupNorm(cout, cout, affine=True)


class _FeatureExtractor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        layers, cin = [], 1
        for i, (co, k, s) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel, cfg.conv_stride)):
            layers.append(_ConvLayer(cin, co, k, s, group_norm=(i == 0)))
            cin = co
        self.conv_layers = nn.ModuleList(layers)


class _FeatureProjection(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer_norm = nn.LayerNorm(cfg.conv_dim[-1], eps=cfg.layer_norm_eps)
        self.projection = nn.Linear(cfg.conv_dim[-1], cfg.hidden_size)


class _WeightNormConv(nn.Module):
    """Conv1d with the old-style ``weight_g`` / ``weight_v`` parameters (weight_norm, dim=2)."""

    def __init__(self, channels: int, k: int, groups: int):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channels))
        v = torch.empty(channels, channels // groups, k)
        nn.init.normal_(v, mean=0, std=2 * (4.0 / (k * channels)) ** 0.5)
        self.weight_g = nn.Parameter(v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())
        self.weight_v = nn.Parameter(v)


class _PosConv(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.conv = _WeightNormConv(cfg.hidden_size, cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups)


class _SelfAttention(nn.Module):
    def __init__(self, h: int):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(h, h) for _ in range(4))


class _FeedForward(nn.Module):
    def __init__(self, h: int, inner: int):
        super().__init__()
        self.intermediate_dense = nn.Linear(h, inner)
        self.output_dense = nn.Linear(inner, h)


class _EncoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = _SelfAttention(cfg.hidden_size)
        self.layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.feed_forward = _FeedForward(cfg.hidden_size, cfg.intermediate_size)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.pos_conv_embed = _PosConv(cfg)
        self.layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.layers = nn.ModuleList([_EncoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])


def _check_config(cfg) -> None:
    bad = []
    if getattr(cfg, "feat_extract_norm", "group") != "group":
        bad.append("feat_extract_norm must be 'group'")
    if getattr(cfg, "do_stable_layer_norm", False):
        bad.append("do_stable_layer_norm must be False")
    if getattr(cfg, "conv_bias", False):
        bad.append("conv_bias must be False")
    if tuple(cfg.conv_stride) != (5, 2, 2, 2, 2, 2, 2) or tuple(cfg.conv_dim) != (512,) * 7:
        bad.append("conv stack must be the wav2vec2-base one")
    if cfg.hidden_size != 768 or cfg.num_attention_heads != 12 or cfg.intermediate_size != 3072:
        bad.append("transformer must be 768 wide / 12 heads / 3072 FFN")
    if getattr(cfg, "add_adapter", False):
        bad.append("add_adapter is not supported")
    if bad:
        raise NotImplementedError("audio_config outside the HIP engine's wav2vec2-base support: " + "; ".join(bad))


class ModifiedWav2Vec2Model(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.config = config if config is not None else AudioConfig()
        _check_config(self.config)
        self.masked_spec_embed = nn.Parameter(torch.empty(self.config.hidden_size).uniform_())
        self.feature_extractor = _FeatureExtractor(self.config)
        self.feature_projection = _FeatureProjection(self.config)
        self.encoder = _Encoder(self.config)
        self.adapter = None
        self._owner = None
        self._register_load_state_dict_pre_hook(self._rename_parametrized)

    @staticmethod
    def _rename_parametrized(state_dict, prefix, *args):
        for old, new in (("parametrizations.weight.original0", "weight_g"), ("parametrizations.weight.original1", "weight_v")):
            k = f"{prefix}encoder.pos_conv_embed.conv.{old}"
            if k in state_dict:
                state_dict[f"{prefix}encoder.pos_conv_embed.conv.{new}"] = state_dict.pop(k)

    def forward(self, input_values: Optional[torch.Tensor], attention_mask: Optional[torch.Tensor] = None,
                mask_time_indices=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                num_frames: Optional[int] = None):
        if attention_mask is not None or mask_time_indices is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("the HIP audio encoder implements the eval path SAID.get_audio_embedding uses: "
                                      "no attention_mask / mask_time_indices / attentions / hidden_states outputs")
        if self._owner is None:
            raise _engine.EngineError("ModifiedWav2Vec2Model must be owned by a SAID model (its engine runs the kernels)")
        lhs = self._owner()._get_engine(1, num_frames or 1).audio_encode(input_values, num_frames, apply_proj=False)
        out = Wav2Vec2BaseModelOutput(last_hidden_state=lhs)
        if return_dict is False:
            return (lhs, None)
        return out
