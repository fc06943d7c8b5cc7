"""Audio utilities of the inference CLI boundary.

Counterparts of /root/reference/said/util/audio.py:20-75 (``load_audio``,
``fit_audio_unet``).  WAV decoding uses ``scipy.io.wavfile`` (torchaudio is not
part of this stack); only 16 kHz input is pinned — resampling parity with
``torchaudio.functional.resample`` is SURVEY.md §8(f) item 2.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class FittedWaveform:
    """Fitted waveform using the window"""

    waveform: torch.FloatTensor
    window_size: int


def load_audio(audio_path: str, sampling_rate: int) -> torch.FloatTensor:
    """Load a WAV file as a mono float waveform in [-1, 1] at ``sampling_rate``."""
    from scipy.io import wavfile

    sr, data = wavfile.read(audio_path)
    if data.dtype == np.int16:
        wav = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        wav = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        wav = (data.astype(np.float32) - 128.0) / 128.0
    else:
        wav = data.astype(np.float32)
    wav_t = torch.from_numpy(wav)
    if wav_t.dim() == 2:  # (frames, channels) -> mean over channels
        wav_t = wav_t.mean(dim=1)
    if sr != sampling_rate:
        raise NotImplementedError(
            f"{audio_path}: sample rate {sr} != {sampling_rate}; resample the file first "
            "(torchaudio-compatible resampling is not part of this build yet)")
    return wav_t


def fit_audio_unet(waveform: torch.FloatTensor, sampling_rate: int, fps: int, divisor_unet: int) -> FittedWaveform:
    """Zero-pad the waveform so the frame count divides ``divisor_unet``."""
    gcd = math.gcd(sampling_rate, fps)
    divisor_waveform = sampling_rate // gcd * divisor_unet
    waveform_len = waveform.shape[0]
    window_len = int(waveform_len / sampling_rate * fps)
    waveform_len_fit = math.ceil(waveform_len / divisor_waveform) * divisor_waveform
    if waveform_len_fit > waveform_len:
        tmp = torch.zeros(waveform_len_fit)
        tmp[:waveform_len] = waveform[:]
        waveform = tmp
    return FittedWaveform(waveform=waveform, window_size=window_len)
