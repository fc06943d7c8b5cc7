"""CPU restatement of the SAID pipeline — TEST INFRASTRUCTURE.

``SAID.process_audio`` / ``get_audio_embedding`` / ``inference``
(/root/reference/said/model/diffusion.py:188-230, 308-472), ``fit_audio_unet``
(said/util/audio.py:42-75) and the CSV writer/reader
(said/util/blendshape.py:36-69).  Random draws are *injected* (initial latents,
editing noise, per-step eta noise) so CPU and GPU paths consume identical noise.
"""
from __future__ import annotations

import csv
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import scheduler as osched
from . import unet as ounet
from . import wav2vec2 as ow2v

SD = Dict[str, torch.Tensor]
SAMPLING_RATE = 16000


def split_state_dict(sd: SD):
    """SAID state dict -> (audio_encoder sd, denoiser sd, null_cond_emb)."""
    a = {k[len("audio_encoder."):]: v for k, v in sd.items() if k.startswith("audio_encoder.")}
    d = {k[len("denoiser."):]: v for k, v in sd.items() if k.startswith("denoiser.")}
    return a, d, sd["null_cond_emb"]


def process_audio(waveform) -> torch.Tensor:
    """diffusion.py:188-207 → HF Wav2Vec2FeatureExtractor(do_normalize=True):
    per-utterance ``(x - mean) / sqrt(var + 1e-7)`` in numpy float32."""
    if isinstance(waveform, torch.Tensor):
        waveform = waveform.numpy()
    if isinstance(waveform, np.ndarray) and waveform.ndim == 1:
        waveform = [waveform]
    out = []
    for w in waveform:
        w = np.asarray(w, dtype=np.float32)
        out.append((w - w.mean()) / np.sqrt(w.var() + 1e-7))
    return torch.from_numpy(np.stack(out).astype(np.float32))


def fit_audio_unet(waveform: torch.Tensor, sampling_rate: int, fps: int, divisor_unet: int):
    """said/util/audio.py:42-75 → (padded waveform, window_len)."""
    gcd = math.gcd(sampling_rate, fps)
    divisor_waveform = sampling_rate // gcd * divisor_unet
    n = waveform.shape[0]
    window_len = int(n / sampling_rate * fps)
    n_fit = math.ceil(n / divisor_waveform) * divisor_waveform
    if n_fit > n:
        tmp = torch.zeros(n_fit)
        tmp[:n] = waveform[:]
        waveform = tmp
    return waveform, window_len


def resample_direct(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> np.ndarray:
    """Direct-form restatement of torchaudio.functional.resample (sinc_interp_hann, default parameters) as called at
    said/util/audio.py:36-37: output sample m sits at time m/new_freq and is the Hann-windowed-sinc weighted sum of the
    input samples within `lowpass_filter_width` zero crossings of the (rolled-off) lower Nyquist rate.  One output at a
    time in float64 — no polyphase bank, no strided convolution — so that it shares nothing with the product's fast
    formulation except the published formula.  torchaudio itself is absent here: parity unpinned.
    """
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    cutoff = min(orig, new) * rolloff            # in units where the input rate is `orig`
    width = math.ceil(lowpass_filter_width * orig / cutoff)
    n = x.shape[-1]
    m_total = math.ceil(new * n / orig)
    y = np.zeros(m_total, dtype=np.float64)
    xd = x.astype(np.float64)
    for m in range(m_total):
        j, i = divmod(m, new)                    # frame and phase
        # taps k = -width .. width + orig - 1 relative to input index j*orig (zero outside the signal)
        k = np.arange(-width, width + orig)
        idx = j * orig + k
        ok = (idx >= 0) & (idx < n)
        t = (k / orig - np.float64(np.float32(i / new))) * cutoff     # the phase offset is a float32 quotient upstream
        t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
        win = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        tp = t * math.pi
        with np.errstate(invalid="ignore", divide="ignore"):
            h = np.where(tp == 0, 1.0, np.sin(tp) / tp) * win * (cutoff / orig)
        h = h.astype(np.float32).astype(np.float64)                      # the bank is stored in float32
        y[m] = np.dot(xd[idx[ok]], h[ok])
    return y.astype(np.float32)


def get_audio_embedding(sd_audio: SD, waveform: torch.Tensor, num_frames: Optional[int]) -> torch.Tensor:
    """diffusion.py:209-230 (feature_dim <= 0: no projection)."""
    return ow2v.wav2vec2_forward(sd_audio, waveform, num_frames)[0]


@dataclass
class OracleOutput:
    result: torch.Tensor
    intermediates: List[torch.Tensor] = field(default_factory=list)


def inference(sd: SD, waveform_processed: torch.Tensor, *, init_latents: torch.Tensor,
              init_samples: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
              edit_noise: Optional[torch.Tensor] = None, step_noise: Optional[Sequence[torch.Tensor]] = None,
              num_inference_steps: int = 100, strength: float = 1.0, guidance_scale: float = 2.5,
              guidance_rescale: float = 0.0, eta: float = 0.0, fps: int = 60, prediction_type: str = "epsilon",
              latent_scale: float = 1.0, save_intermediate: bool = False,
              audio_embedding: Optional[torch.Tensor] = None) -> OracleOutput:
    """diffusion.py:354-472.  ``init_latents`` replaces the ``randn`` of :363-367
    when ``init_samples`` is None; ``edit_noise`` replaces the ``randn`` inside
    ``add_noise`` (:377-385); ``step_noise[k]`` is the eta>0 variance noise."""
    sd_audio, sd_unet, null_cond = split_state_dict(sd)
    B, Ta = waveform_processed.shape
    do_cfg = guidance_scale > 1.0
    window = int(Ta / SAMPLING_RATE * fps)
    sch = osched.OracleDDIM(1000, prediction_type)
    sch.set_timesteps(num_inference_steps)

    latents = init_latents.clone() if init_samples is None else init_samples.clone()
    latents = latents * (latent_scale * sch.init_noise_sigma)
    init_lat = latents.clone()
    init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
    noise = None
    if init_samples is not None:
        t0 = sch.timesteps[-init_timestep]
        noise = edit_noise
        latents = sch.add_noise(latents, noise, torch.tensor([int(t0)] * B, dtype=torch.long))

    if audio_embedding is None:
        audio_embedding = get_audio_embedding(sd_audio, waveform_processed, window)
    if do_cfg:
        uncond = null_cond.repeat(B, audio_embedding.shape[1], 1)
        audio_embedding = torch.cat([uncond, audio_embedding])

    inter: List[torch.Tensor] = []
    t_start = num_inference_steps - init_timestep
    for idx, t in enumerate(sch.timesteps[t_start:]):
        if save_intermediate:
            inter.append((latents / latent_scale).clone())
        x = torch.cat([latents] * 2) if do_cfg else latents
        tt = t.repeat(x.shape[0])
        pred = ounet.unet1d_forward(sd_unet, x, tt, audio_embedding)
        if do_cfg:
            e_u, e_c = pred.chunk(2)
            pred = e_c + guidance_scale * (e_c - e_u)
            if guidance_rescale > 0.0:
                pred = osched.rescale_noise_cfg(pred, e_c, guidance_rescale)
        vn = step_noise[idx] if (eta > 0 and step_noise is not None) else None
        latents = sch.step(pred, int(t), latents, eta=eta, variance_noise=vn)
        if init_samples is not None and mask is not None:
            noisy = init_lat
            nxt = t_start + idx + 1
            if nxt < num_inference_steps:
                noisy = sch.add_noise(init_lat, noise, sch.timesteps[nxt])
            latents = noisy * mask + latents * (1 - mask)
    result = (latents / latent_scale).clamp(0, 1)
    return OracleOutput(result=result, intermediates=inter)


# ---- CSV I/O (said/util/blendshape.py:36-69; header = dataset_voca.py:99-132) ----
BLENDSHAPE_CLASSES = [
    "jawForward", "jawLeft", "jawRight", "jawOpen", "mouthClose", "mouthFunnel", "mouthPucker", "mouthLeft",
    "mouthRight", "mouthSmileLeft", "mouthSmileRight", "mouthFrownLeft", "mouthFrownRight", "mouthDimpleLeft",
    "mouthDimpleRight", "mouthStretchLeft", "mouthStretchRight", "mouthRollLower", "mouthRollUpper",
    "mouthShrugLower", "mouthShrugUpper", "mouthPressLeft", "mouthPressRight", "mouthLowerDownLeft",
    "mouthLowerDownRight", "mouthUpperUpLeft", "mouthUpperUpRight", "cheekPuff", "cheekSquintLeft",
    "cheekSquintRight", "noseSneerLeft", "noseSneerRight",
]


def load_blendshape_coeffs(path: str) -> torch.Tensor:
    with open(path, newline="") as f:
        rows = list(csv.reader(f))
    return torch.tensor([[float(v) for v in r] for r in rows[1:]], dtype=torch.float32)
