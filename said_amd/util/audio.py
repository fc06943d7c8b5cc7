"""Audio ingest of the inference CLI boundary: WAV -> mono float waveform at the model's rate, padded for the UNet.

Counterparts of the reference's `load_audio` and `fit_audio_unet` (/root/reference/said/util/audio.py:20-39, 42-75).
The reference decodes and resamples with torchaudio, which is not part of this stack: WAV decoding uses
`scipy.io.wavfile` with torchaudio's integer normalisation, and `resample` restates torchaudio's published
`functional.resample` algorithm (band-limited interpolation with a Hann-windowed sinc, defaults
lowpass_filter_width=6, rolloff=0.99).  torchaudio is absent from /root/reference and from this image, so that
restatement is **parity-unpinned**; tests check it against an independent direct-form evaluation and against
signal-level properties (length rule, DC gain, tone preservation).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class FittedWaveform:
    """A waveform zero-padded for the UNet and the number of coefficient frames the original audio covers."""

    waveform: torch.FloatTensor
    window_size: int


def _sinc_filter_bank(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Polyphase filter bank (new, 1, taps) of the Hann-windowed sinc interpolator and its half width.

    `orig`, `new`: the two rates divided by their gcd.  Phase i produces output sample j*new + i from the input
    samples around j*orig; times are measured in units of the lower of the two rates times `rolloff`.
    """
    cutoff = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / cutoff)
    taps = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    phase = torch.arange(0, -new, -1)[:, None, None] / new          # int64 / int -> float32, as in torchaudio
    t = (phase + taps) * cutoff
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (cutoff / orig)
    return kernel.to(torch.float32), width


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """Band-limited resampling of (..., time) float32 audio from `orig_freq` to `new_freq` Hz."""
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kernel, width = _sinc_filter_bank(orig, new)
    lead = waveform.shape[:-1]
    flat = waveform.reshape(-1, waveform.shape[-1]).to(torch.float32)
    length = flat.shape[-1]
    padded = torch.nn.functional.pad(flat, (width, width + orig))
    frames = torch.nn.functional.conv1d(padded[:, None], kernel, stride=orig)    # (n, new, frames)
    out = frames.transpose(1, 2).reshape(flat.shape[0], -1)
    out = out[:, : math.ceil(new * length / orig)]
    return out.reshape(*lead, out.shape[-1])


def _decode_wav(path: str):
    """(channels, time) float32 in [-1, 1) and the file's sample rate (integer PCM scaled by 2**-(bits-1))."""
    from scipy.io import wavfile

    rate, data = wavfile.read(path)
    if data.dtype == np.uint8:
        pcm = (data.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(data.dtype, np.integer):
        pcm = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
    else:
        pcm = data.astype(np.float32)
    pcm = pcm[:, None] if pcm.ndim == 1 else pcm
    return torch.from_numpy(np.ascontiguousarray(pcm.T)), int(rate)


def _decode(path: str):
    """(channels, time) float32 and the file's sample rate.  RIFF/WAV through scipy (always present); other containers (flac, ogg, ...: the reference's
    `torchaudio.load` takes any, audio.py:34) through `soundfile` when that package is importable — neither it nor torchaudio ships in this stack, so
    there the error says what is missing instead of guessing."""
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic in (b"RIFF", b"RIFX", b"RF64"):
        return _decode_wav(path)
    try:
        import soundfile as sf
    except ImportError as e:
        raise RuntimeError(f"{path}: not a RIFF/WAV file, and decoding other containers needs the `soundfile` package (not installed); "
                           "convert the audio to WAV") from e
    data, rate = sf.read(path, dtype="float32", always_2d=True)     # (time, channels), integer PCM scaled to [-1, 1) like torchaudio
    return torch.from_numpy(np.ascontiguousarray(data.T)), int(rate)


def load_audio(audio_path: str, sampling_rate: int) -> torch.FloatTensor:
    """Mono waveform (T_a,) at `sampling_rate`: decode, resample each channel if needed, average the channels."""
    channels, rate = _decode(audio_path)
    if rate != sampling_rate:
        channels = resample(channels, rate, sampling_rate)
    return channels.mean(dim=0)


def fit_audio_unet(waveform: torch.FloatTensor, sampling_rate: int, fps: int, divisor_unet: int) -> FittedWaveform:
    """Zero-pad at the end so that the audio spans a whole number of `divisor_unet`-frame groups.

    One coefficient frame lasts sampling_rate/fps samples; in lowest terms that is `sampling_rate // gcd` samples per
    `fps // gcd` frames, and the reference pads to a multiple of `sampling_rate // gcd * divisor_unet` samples
    (audio.py:64-73).  `window_size` is the frame count of the UN-padded audio, floor(len / sampling_rate * fps).
    """
    quantum = sampling_rate // math.gcd(sampling_rate, fps) * divisor_unet
    n = int(waveform.shape[0])
    missing = -n % quantum
    padded = torch.nn.functional.pad(waveform, (0, missing)) if missing else waveform
    return FittedWaveform(waveform=padded, window_size=int(n / sampling_rate * fps))
