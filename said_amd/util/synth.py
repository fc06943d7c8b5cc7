"""Deterministic synthetic weights and inputs.

No pretrained SAiD.pth / wav2vec2 checkpoint is reachable offline, and the real
files (28 MB + 377 MB) are too large to commit, so tests, goldens and bench.py
all use a *name-keyed* fill that every party can regenerate bit-identically on
the CPU: ``torch.Generator().manual_seed(crc32(name))`` → ``randn`` × scale.

The reference zero-initialises several layers (``zero_module`` at
said/model/ldm/openaimodel.py:182-184,668 and attention.py:221), which would
make an untouched random model output exactly 0; the fill re-randomises them.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import torch

# ---------------------------------------------------------------------------
# UNet1D denoiser (reference: said/model/unet_1d_condition.py:36-49)
# ---------------------------------------------------------------------------
MODEL_CH = 192
TIME_CH = 768
HEADS = 6
HEAD_DIM = 32
FF_INNER = 768


def unet_param_shapes(in_ch: int = 32, out_ch: int = 32, ctx_dim: int = 768) -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of ``UNet1DConditionModel.state_dict()`` (160 tensors)."""
    C, E = MODEL_CH, TIME_CH
    s: Dict[str, Tuple[int, ...]] = {}
    s["model.time_embed.0.weight"] = (E, C)
    s["model.time_embed.0.bias"] = (E,)
    s["model.time_embed.2.weight"] = (E, E)
    s["model.time_embed.2.bias"] = (E,)
    s["model.input_blocks.0.0.weight"] = (C, in_ch, 3)
    s["model.input_blocks.0.0.bias"] = (C,)

    def res(p: str, cin: int) -> None:
        s[p + ".in_layers.0.weight"] = (cin,)
        s[p + ".in_layers.0.bias"] = (cin,)
        s[p + ".in_layers.2.weight"] = (C, cin, 3)
        s[p + ".in_layers.2.bias"] = (C,)
        s[p + ".emb_layers.1.weight"] = (C, E)
        s[p + ".emb_layers.1.bias"] = (C,)
        s[p + ".out_layers.0.weight"] = (C,)
        s[p + ".out_layers.0.bias"] = (C,)
        s[p + ".out_layers.3.weight"] = (C, C, 3)
        s[p + ".out_layers.3.bias"] = (C,)
        if cin != C:
            s[p + ".skip_connection.weight"] = (C, cin, 1)
            s[p + ".skip_connection.bias"] = (C,)

    def st(p: str) -> None:
        s[p + ".norm.weight"] = (C,)
        s[p + ".norm.bias"] = (C,)
        b = p + ".transformer_blocks.0"
        for a, kd in (("attn1", C), ("attn2", ctx_dim)):
            s[f"{b}.{a}.to_q.weight"] = (C, C)
            s[f"{b}.{a}.to_k.weight"] = (C, kd)
            s[f"{b}.{a}.to_v.weight"] = (C, kd)
            s[f"{b}.{a}.to_out.0.weight"] = (C, C)
            s[f"{b}.{a}.to_out.0.bias"] = (C,)
        s[f"{b}.ff.net.0.proj.weight"] = (2 * FF_INNER, C)
        s[f"{b}.ff.net.0.proj.bias"] = (2 * FF_INNER,)
        s[f"{b}.ff.net.2.weight"] = (C, FF_INNER)
        s[f"{b}.ff.net.2.bias"] = (C,)
        for n in ("norm1", "norm2", "norm3"):
            s[f"{b}.{n}.weight"] = (C,)
            s[f"{b}.{n}.bias"] = (C,)
        s[p + ".proj_out.weight"] = (C, C, 1)
        s[p + ".proj_out.bias"] = (C,)

    res("model.input_blocks.1.0", C)
    st("model.input_blocks.1.1")
    res("model.middle_block.0", C)
    st("model.middle_block.1")
    res("model.middle_block.2", C)
    res("model.output_blocks.0.0", 2 * C)
    st("model.output_blocks.0.1")
    res("model.output_blocks.1.0", 2 * C)
    st("model.output_blocks.1.1")
    s["model.out.0.weight"] = (C,)
    s["model.out.0.bias"] = (C,)
    s["model.out.2.weight"] = (out_ch, C, 3)
    s["model.out.2.bias"] = (out_ch,)
    return s


# ---------------------------------------------------------------------------
# Wav2Vec2 base encoder (default Wav2Vec2Config(); reference diffusion.py:86-89)
# ---------------------------------------------------------------------------
W2V_CONV_DIM = (512,) * 7
W2V_CONV_KERNEL = (10, 3, 3, 3, 3, 2, 2)
W2V_CONV_STRIDE = (5, 2, 2, 2, 2, 2, 2)
W2V_HIDDEN = 768
W2V_LAYERS = 12
W2V_HEADS = 12
W2V_FFN = 3072
W2V_POS_K = 128
W2V_POS_GROUPS = 16


def w2v_param_shapes(num_layers: int = W2V_LAYERS) -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of the audio encoder state dict in the transformers-4.30.2
    naming the reference checkpoint uses (``…pos_conv_embed.conv.weight_g/_v``)."""
    H = W2V_HIDDEN
    s: Dict[str, Tuple[int, ...]] = {}
    s["masked_spec_embed"] = (H,)
    cin = 1
    for i, (co, k) in enumerate(zip(W2V_CONV_DIM, W2V_CONV_KERNEL)):
        s[f"feature_extractor.conv_layers.{i}.conv.weight"] = (co, cin, k)
        if i == 0:
            s["feature_extractor.conv_layers.0.layer_norm.weight"] = (co,)
            s["feature_extractor.conv_layers.0.layer_norm.bias"] = (co,)
        cin = co
    s["feature_projection.layer_norm.weight"] = (cin,)
    s["feature_projection.layer_norm.bias"] = (cin,)
    s["feature_projection.projection.weight"] = (H, cin)
    s["feature_projection.projection.bias"] = (H,)
    s["encoder.pos_conv_embed.conv.bias"] = (H,)
    s["encoder.pos_conv_embed.conv.weight_g"] = (1, 1, W2V_POS_K)
    s["encoder.pos_conv_embed.conv.weight_v"] = (H, H // W2V_POS_GROUPS, W2V_POS_K)
    s["encoder.layer_norm.weight"] = (H,)
    s["encoder.layer_norm.bias"] = (H,)
    for l in range(num_layers):
        p = f"encoder.layers.{l}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[f"{p}.attention.{n}.weight"] = (H, H)
            s[f"{p}.attention.{n}.bias"] = (H,)
        s[f"{p}.layer_norm.weight"] = (H,)
        s[f"{p}.layer_norm.bias"] = (H,)
        s[f"{p}.feed_forward.intermediate_dense.weight"] = (W2V_FFN, H)
        s[f"{p}.feed_forward.intermediate_dense.bias"] = (W2V_FFN,)
        s[f"{p}.feed_forward.output_dense.weight"] = (H, W2V_FFN)
        s[f"{p}.feed_forward.output_dense.bias"] = (H,)
        s[f"{p}.final_layer_norm.weight"] = (H,)
        s[f"{p}.final_layer_norm.bias"] = (H,)
    return s


def vae_encoder_param_shapes() -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of the ``encoder.*`` half of the reference's BCVAE state dict (said/model/vae.py:26-64;
    matches model/vae.pth), BatchNorm running statistics included."""
    s: Dict[str, Tuple[int, ...]] = {}
    for idx, (co, ci, k) in {0: (32, 32, 3), 3: (64, 32, 3), 6: (64, 64, 4), 9: (32, 64, 3)}.items():
        s[f"encoder.conv_layers.{idx}.weight"] = (co, ci, k)
        s[f"encoder.conv_layers.{idx}.bias"] = (co,)
    for idx, n in {1: 32, 4: 64, 7: 64}.items():
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"encoder.conv_layers.{idx}.{leaf}"] = (n,)
    for idx, (co, ci) in {0: (256, 1760), 3: (128, 256), 6: (64, 128)}.items():
        s[f"encoder.fc_layers.{idx}.weight"] = (co, ci)
        s[f"encoder.fc_layers.{idx}.bias"] = (co,)
    for idx, n in {1: 256, 4: 128}.items():
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"encoder.fc_layers.{idx}.{leaf}"] = (n,)
    for nme in ("fc_mu", "fc_logvar"):
        s[f"encoder.{nme}.weight"] = (64, 64)
        s[f"encoder.{nme}.bias"] = (64,)
    return s


_VAE_BN = ("encoder.conv_layers.1.", "encoder.conv_layers.4.", "encoder.conv_layers.7.", "encoder.fc_layers.1.", "encoder.fc_layers.4.")


def vae_encoder_state_dict(salt: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic fill of the BCVAE encoder: BatchNorm affine ~ (1, 0) +- 0.1, running_mean ~ 0.1 N(0,1),
    running_var in [0.5, 1.5]; conv / linear weights variance-preserving."""
    sd = {}
    for k, shp in vae_encoder_param_shapes().items():
        t = fill_tensor(k, shp, salt)
        if k.startswith(_VAE_BN):
            g = torch.Generator()
            g.manual_seed((zlib.crc32(k.encode()) + 7919 * salt) & 0x7FFFFFFF)
            r = torch.randn(shp, generator=g, dtype=torch.float32)
            leaf = k.split(".")[-1]
            t = {"weight": 1.0 + 0.1 * r, "bias": 0.1 * r, "running_mean": 0.1 * r, "running_var": 1.0 + 0.5 * torch.tanh(r)}[leaf]
        sd[k] = t
    return sd


# ---------------------------------------------------------------------------
# Deterministic fill
# ---------------------------------------------------------------------------
def _is_norm_name(name: str) -> bool:
    parts = name.split(".")
    leaf_parent = parts[-2] if len(parts) >= 2 else ""
    if leaf_parent in ("norm", "norm1", "norm2", "norm3", "layer_norm", "final_layer_norm"):
        return True
    # GroupNorm32 inside nn.Sequential: in_layers.0 / out_layers.0 / out.0
    if leaf_parent == "0" and len(parts) >= 3 and parts[-3] in ("in_layers", "out_layers", "out"):
        return True
    return False


def fill_tensor(name: str, shape: Tuple[int, ...], salt: int = 0) -> torch.Tensor:
    """Bit-reproducible CPU fill for one named parameter."""
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    leaf = name.split(".")[-1]
    if _is_norm_name(name):
        return 1.0 + 0.1 * r if leaf == "weight" else 0.1 * r
    if leaf == "weight_g":  # weight-norm magnitude: keep positive, O(1)
        return 1.0 + 0.1 * r.abs()
    if leaf == "bias":
        return 0.02 * r
    if name in ("null_cond_emb", "masked_spec_embed"):
        return r
    # linear / conv weights: variance-preserving 1/sqrt(fan_in)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return r * (1.0 / max(fan_in, 1)) ** 0.5


def fill_state_dict(shapes: Dict[str, Tuple[int, ...]], prefix: str = "", salt: int = 0) -> Dict[str, torch.Tensor]:
    """Values are keyed by the *unprefixed* name, so ``denoiser.model.x`` inside a full SAID
    state dict equals ``model.x`` of a standalone UNet fill (the golden vectors' weights)."""
    return {prefix + k: fill_tensor(k, shp, salt) for k, shp in shapes.items()}


def said_state_dict(num_w2v_layers: int = W2V_LAYERS, ctx_dim: int = 768, salt: int = 0) -> Dict[str, torch.Tensor]:
    """Full ``SAID_UNet1D.state_dict()`` look-alike (reference key layout,
    SURVEY.md §8b): ``null_cond_emb``, ``audio_encoder.*``, ``denoiser.*``."""
    sd = {"null_cond_emb": fill_tensor("null_cond_emb", (1, 1, ctx_dim), salt)}
    sd.update(fill_state_dict(w2v_param_shapes(num_w2v_layers), "audio_encoder.", salt))
    sd.update(fill_state_dict(unet_param_shapes(32, 32, ctx_dim), "denoiser.", salt))
    return sd


def state_dict_checksum(sd: Dict[str, torch.Tensor], names: Iterable[str] | None = None) -> float:
    """Order-independent fp64 checksum used to pin fixtures to a fill."""
    tot = 0.0
    for k in (names if names is not None else sorted(sd)):
        v = sd[k].double()
        tot += float(v.sum()) + 0.5 * float((v * v).sum())
    return tot


def synth_waveform(index: int, num_samples: int) -> torch.Tensor:
    """SURVEY.md §8d synthetic audio: ``randn(Ta) * 0.1`` seeded 1234+index."""
    g = torch.Generator()
    g.manual_seed(1234 + index)
    return torch.randn(num_samples, generator=g, dtype=torch.float32) * 0.1


def synth_latents(index: int, shape: Tuple[int, ...]) -> torch.Tensor:
    """Host-generated initial latents / per-step noise, seeded by ``index``."""
    g = torch.Generator()
    g.manual_seed(index)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def trained_like_state_dict(num_w2v_layers: int = W2V_LAYERS, ctx_dim: int = 768, salt: int = 0) -> Dict[str, torch.Tensor]:
    """said_state_dict() reshaped towards what trained checkpoints look like (none is reachable offline): the denoiser's linear / conv
    weights get a log-normal heavy tail (x exp(0.5 N(0, 1))), two outlier output rows and two outlier input columns per matrix (x 8), and
    a tenth of every GroupNorm / LayerNorm's gains are drawn log-uniformly from [0.1, 10] with biases up to +-1 — the statistics that
    stress reduced-precision products (outlier channels, gains >> 1, un-normalised residual streams of O(100)).  Deterministic, name-keyed."""
    sd = said_state_dict(num_w2v_layers, ctx_dim, salt)
    for k in list(sd):
        if not k.startswith("denoiser."):
            continue
        v = sd[k]
        g = torch.Generator()
        g.manual_seed((zlib.crc32(("trained:" + k).encode()) + 7919 * salt) & 0x7FFFFFFF)
        name = k[len("denoiser."):]
        if _is_norm_name(name):
            n = v.shape[0]
            pick = torch.rand(n, generator=g) < 0.1
            if k.endswith(".weight"):
                wide = torch.exp((torch.rand(n, generator=g) * 2 - 1) * 2.302585092994046)   # log-uniform in [0.1, 10]
                v = torch.where(pick, wide, v)
            else:
                v = torch.where(pick, torch.rand(n, generator=g) * 2 - 1, v)
        elif v.dim() >= 2:
            v = v * torch.exp(0.5 * torch.randn(v.shape, generator=g))
            rows = torch.randint(0, v.shape[0], (2,), generator=g)
            cols = torch.randint(0, v.shape[1], (2,), generator=g)
            v = v.clone()
            v[rows] *= 8.0
            v[:, cols] *= 8.0
        sd[k] = v.contiguous()
    return sd
