"""Offline stand-in for ``Wav2Vec2Processor.from_pretrained("facebook/wav2vec2-base-960h")``
as used by ``SAID.process_audio`` (/root/reference/said/model/diffusion.py:90-95, 188-207).

Only the feature-extractor half is on the path: per-utterance zero-mean /
unit-variance normalisation in numpy float32, ``(x - mean) / sqrt(var + 1e-7)``
(HF ``Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm``, ``do_normalize=True``,
``return_attention_mask=False`` for that checkpoint).  This is host-side input
preparation on the CPU in the reference as well (the caller moves the result to
the device, script/inference.py:171).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Union

import numpy as np
import torch


class AudioProcessor:
    def __init__(self, sampling_rate: int = 16000):
        self.feature_extractor = SimpleNamespace(sampling_rate=sampling_rate, do_normalize=True, padding_value=0.0)

    def __call__(self, raw_speech: Union[np.ndarray, torch.Tensor, List[np.ndarray]], sampling_rate: int = None,
                 return_tensors: str = "pt", **kwargs):
        if sampling_rate is not None and sampling_rate != self.feature_extractor.sampling_rate:
            raise ValueError(f"The model was trained at {self.feature_extractor.sampling_rate} Hz, got {sampling_rate}")
        if isinstance(raw_speech, torch.Tensor):
            raw_speech = raw_speech.detach().cpu().numpy()
        if isinstance(raw_speech, np.ndarray) and raw_speech.ndim == 1:
            raw_speech = [raw_speech]
        arrs = [np.asarray(a, dtype=np.float32) for a in raw_speech]
        if len({a.shape[0] for a in arrs}) != 1:
            raise ValueError("all utterances in a batch must have the same length (no padding on this path)")
        normed = [(a - a.mean()) / np.sqrt(a.var() + 1e-7) for a in arrs]
        vals = np.stack(normed).astype(np.float32)
        return {"input_values": torch.from_numpy(vals) if return_tensors == "pt" else vals}
