"""SAiD diffusion pipeline on the MI355X engine.

Drop-in for the inference surface of /root/reference/said/model/diffusion.py:
``SAIDInferenceOutput``, ``SAIDNoiseAdditionOutput``, ``SAID``, ``SAID_UNet1D``
with the reference's constructor / method signatures and ``state_dict()`` key
layout (``null_cond_emb``, ``audio_encoder.*``, ``denoiser.model.*``, optional
``audio_proj_layer.*``).  Host code only owns tensors and tables; all device
math is in hand-written HIP kernels (include/said_hip.h).  No CPU fallback.

Extensions that do not change the reference signatures (keyword-only, default
off): ``inference(..., init_latents=, edit_noise=, step_noise=)`` inject the
three random draws of the reference (``torch.randn`` at diffusion.py:364, inside
``add_noise`` at :383-385, and per-step inside ``scheduler.step`` for eta > 0)
so a CPU oracle and the GPU path can be fed identical noise.
"""
from __future__ import annotations

import threading
import warnings
import weakref
from abc import ABC
from dataclasses import dataclass
from typing import List, Optional, Type, Union

import numpy as np
import torch
from torch import nn

from .. import _engine
from ..scheduler import DDIMScheduler
from .processor import AudioProcessor
from .unet_1d_condition import UNet1DConditionModel
from .wav2vec2 import AudioConfig, ModifiedWav2Vec2Model


@dataclass
class SAIDInferenceOutput:
    """Dataclass for the inference output"""

    result: torch.FloatTensor  # (Batch_size, sample_seq_len, x_dim), generated blendshape coefficients
    intermediates: List[torch.FloatTensor]  # (Batch_size, sample_seq_len, x_dim), pre-step latents


@dataclass
class SAIDNoiseAdditionOutput:
    """Dataclass for the noise addition output"""

    noisy_sample: torch.FloatTensor
    noise: torch.FloatTensor
    velocity: torch.FloatTensor



class _Progress:
    """`show_process=True` (diffusion.py:412-415: tqdm around the reference's Python loop).  Here the loop is a queue of hipGraph launches
    the host does not step through, so a thread polls the engine's device-side step counter (said_loop_progress: a private stream,
    the loop's stream is never blocked) and drives the same kind of bar on stderr."""

    def __init__(self, eng, total: int, interval: float = 0.05):
        from tqdm import tqdm
        self._eng, self._total, self._interval = eng, total, interval
        eng.loop_progress_reset()   # the previous loop's final count must not be read as this loop's progress
        self._bar = tqdm(total=total)
        self._done = 0
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._poll, name="said-show-process", daemon=True)
        self._th.start()

    def _advance(self):
        try:
            s = min(max(self._eng.loop_progress(), 0), self._total)
        except Exception:   # the engine was closed under us: nothing to show
            return
        if s > self._done:
            self._bar.update(s - self._done)
            self._done = s

    def _poll(self):
        while not self._stop.wait(self._interval):
            self._advance()

    def close(self):
        self._stop.set()
        self._th.join()
        self._advance()
        self._bar.close()

class SAID(ABC, nn.Module):
    """Abstract class of SAiD models"""

    denoiser: nn.Module

    def __init__(self, audio_config=None, audio_processor=None, noise_scheduler: Type = DDIMScheduler, in_channels: int = 32,
                 feature_dim: int = -1, diffusion_steps: int = 1000, latent_scale: float = 1,
                 prediction_type: str = "epsilon"):
        super().__init__()
        # Audio-related
        self.audio_config = audio_config if audio_config is not None else AudioConfig()
        self.audio_encoder = ModifiedWav2Vec2Model(self.audio_config)
        # The reference downloads facebook/wav2vec2-base-960h's processor here; its only effect on
        # this path is zero-mean/unit-variance normalisation, provided offline by AudioProcessor.
        self.audio_processor = audio_processor if audio_processor is not None else AudioProcessor(16000)
        self.sampling_rate = self.audio_processor.feature_extractor.sampling_rate
        self.latent_scale = latent_scale
        # Noise scheduler
        self.noise_scheduler = noise_scheduler(num_train_timesteps=diffusion_steps, beta_schedule="squaredcos_cap_v2",
                                               prediction_type=prediction_type)
        # Feature embedding
        self.feature_dim = feature_dim
        hidden = self.audio_config.output_hidden_size
        if self.feature_dim > 0:
            self.audio_proj_layer = nn.Linear(hidden, self.feature_dim)
            self.null_cond_emb = nn.Parameter(torch.randn(1, 1, self.feature_dim))
        else:
            self.null_cond_emb = nn.Parameter(torch.randn(1, 1, hidden))
        self._eng: Optional[_engine.Engine] = None
        self._eng_key = None
        self._eng_stale = True
        self._param_list = None
        self.mfma_dtype = "fp32"   # "fp32" (split-fp16 products) | "fp32_strict" (fp32 matrix instructions) | "bf16" (BASELINE.json configs[2]); set_mfma_dtype()
        # What inference() / forward() do when the model output of a step is not finite (said_numeric_status).  In "fp32" mode that can be an operand beyond the
        # split-fp16 products' domain (|x| < 65504: include/said_hip.h) where the reference's fp32 would have been fine:
        #   "strict_retry" (default): warn and run the call again on fp32 matrix instructions, with the same random draws;
        #   "raise": EngineError;   "ignore": no check (and no stream synchronisation at the end of inference()).
        self.on_nonfinite = "strict_retry"
        self.dedupe_audio = True    # SAID.inference encodes byte-identical rows of a batch once (not part of the reference surface)
        self.clip_groups = None     # None: decided per call (_pick_clip_groups); n >= 1: that many concurrent clip groups
        self._clones: List[_engine.Engine] = []
        self.audio_encoder._owner = weakref.ref(self)

    # ---- engine management ---------------------------------------------------
    # The engine holds its own packed copy of the weights.  It is rebuilt when the module's parameters are replaced or
    # moved — load_state_dict(), .to() / .cuda() / .float() (all go through nn.Module._apply) mark it stale — and when the
    # cheap per-call check below sees an in-place update that bumps a parameter's version counter (optimizer steps,
    # `p.copy_()` under no_grad).  Writes through `p.data` bypass both: call refresh_engine() after them.
    def _apply(self, fn, *args, **kwargs):
        self._eng_stale = True
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._eng_stale = True
        return super().load_state_dict(*args, **kwargs)

    def refresh_engine(self) -> "SAID":
        """Re-upload the weights at the next call (after in-place edits of `param.data`).  Not part of the reference surface."""
        self._eng_stale = True
        return self

    def __setattr__(self, name, value):
        # a parameter or submodule replaced on the model (`model.null_cond_emb = nn.Parameter(..)`, `model.denoiser = ...`): the cached
        # parameter list below would keep pointing at the old tensors
        if isinstance(value, (torch.nn.Parameter, torch.nn.Module)):
            self.__dict__["_eng_stale"] = True
            self.__dict__["_param_list"] = None
        super().__setattr__(name, value)

    def _weights_key(self):
        ps = self._param_list
        if ps is None:
            ps = self._param_list = list(self.parameters())
        # versions: in-place updates; data pointers: `.to()` / `.half()` / `.data = ...` on a SUBMODULE (they replace param.data without
        # bumping the version counter and do not pass through this module's _apply)
        return (str(ps[0].device), sum(p._version for p in ps), sum(p.data_ptr() for p in ps))

    def _get_engine(self, batch_eff: int, frames: int) -> _engine.Engine:
        e = self._eng
        if e is None or getattr(self, "_eng_stale", True) or self._weights_key() != self._eng_key:
            self._param_list = None
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise _engine.EngineError(f"model is on {dev}: said_amd runs on MI355X only — call .to('cuda:N') "
                                          "(the CPU restatement lives in oracle/ and is test infrastructure)")
            cap_b = max(batch_eff, e.max_batch_eff if e else 2)
            cap_t = max((frames + 63) // 64 * 64, e.max_frames if e else 64)
            if e is not None:
                e.close()
            ctx_dim = self.feature_dim if self.feature_dim > 0 else self.audio_config.hidden_size
            e = _engine.Engine(dev, cap_b, cap_t, self.denoiser.in_channels, ctx_dim)
            e.load_weights(self.state_dict())
            self._eng, self._eng_key, self._eng_stale = e, self._weights_key(), False
            self._clones = []      # (closed with the old engine)
            self.noise_scheduler._engine = e
        elif e.max_batch_eff < batch_eff or e.max_frames < frames:
            # a larger batch or a longer clip: only the workspace grows, the packed weights stay where they are
            try:
                e.reserve(max(batch_eff, e.max_batch_eff), max((frames + 63) // 64 * 64, e.max_frames))
            except _engine.EngineError:
                # out of memory while growing: the context has lost its workspace (said_reserve) — drop it, the next call builds a fresh one
                for c in self._clones:
                    c.close()
                self._clones = []
                e.close()
                self._eng = None
                raise
        e.set_precision(self.mfma_dtype)
        return e

    def set_mfma_dtype(self, dtype: str) -> "SAID":
        """How the matrix products multiply (include/said_hip.h, said_set_precision): "fp32" (default: fp32 tensors, products on split-fp16
        operands, 22-bit significands, operand domain |x| < 65504), "fp32_strict" (fp32 matrix instructions on fp32 operands: the reference's
        arithmetic, slower) or "bf16" (operands rounded to bfloat16).  Accumulation and everything else is fp32 in every mode.  Not part of the
        reference surface."""
        if dtype not in _engine.PRECISIONS:
            raise ValueError(f"mfma dtype must be one of {sorted(_engine.PRECISIONS)}, got {dtype!r}")
        self.mfma_dtype = dtype
        return self

    def _nonfinite(self, engines) -> Optional[str]:
        """None, or a description of the first engine whose last call produced an inf / NaN (synchronises the current stream)."""
        for i, e in enumerate(engines):
            step, res = e.numeric_status()
            if step >= 0 or res:
                where = f"model output of denoise step {step}" if step >= 0 else "result"
                return f"{where} is not finite" + (f" (clip group {i})" if len(engines) > 1 else "")
        return None

    # ---- reference surface -----------------------------------------------------
    def forward(self, noisy_samples: torch.FloatTensor, timesteps: torch.LongTensor, audio_embedding: torch.FloatTensor) -> torch.FloatTensor:
        """Predicted noise for (B, T, C) noisy coefficients, timesteps (B,)|(1,)|(), audio tokens (B, S, D)."""
        timestep_size = timesteps.size()
        if len(timestep_size) == 0 or timestep_size[0] == 1:
            timesteps = timesteps.reshape(-1)[:1].repeat(noisy_samples.shape[0])
        out = self.denoiser(noisy_samples, timesteps, audio_embedding)
        if self.on_nonfinite != "ignore" and self._eng is not None:
            bad = self._nonfinite([self._eng])
            if bad is not None and self._eng.effective_precision() == "fp32":
                if self.on_nonfinite == "raise":
                    raise _engine.EngineError(f"SAID.forward: {bad} in fp32 mode (split-fp16 products, operand domain |x| < 65504); use set_mfma_dtype('fp32_strict')")
                warnings.warn(f"SAID.forward: {bad} on split-fp16 products; evaluating again on fp32 matrix instructions (set_mfma_dtype('fp32_strict') avoids the first attempt)",
                              RuntimeWarning, stacklevel=2)
                old, self.mfma_dtype = self.mfma_dtype, "fp32_strict"
                try:
                    out = self.denoiser(noisy_samples, timesteps, audio_embedding)
                finally:
                    self.mfma_dtype = old
        return out

    def pred_original_sample(self, noisy_samples, noise, timesteps):
        """x_0 = (x_t - sqrt(1-ā) eps) / sqrt(ā)  (diffusion.py:157-186)."""
        ac = self.noise_scheduler.alphas_cumprod[torch.as_tensor(timesteps).cpu()].reshape(-1)
        inv = (1.0 / ac ** 0.5).tolist()
        nb = (-(1 - ac) ** 0.5 / ac ** 0.5).tolist()
        B = noisy_samples.shape[0]
        if len(inv) == 1 and B > 1:   # 0-dim / length-1 timesteps broadcast over the batch (reference: .view(-1, 1, 1))
            inv, nb = inv * B, nb * B
        e = self._get_engine(noisy_samples.shape[0], noisy_samples.shape[1])
        return e.axpby(inv, noisy_samples, nb, noise)

    def process_audio(self, waveform: Union[np.ndarray, torch.Tensor, List[np.ndarray]]) -> torch.FloatTensor:
        """(audio_seq_len,) or list thereof → (B, T_a) processed mono waveform (CPU, as in the reference)."""
        return self.audio_processor(waveform, sampling_rate=self.sampling_rate, return_tensors="pt")["input_values"]

    def get_audio_embedding(self, waveform: torch.FloatTensor, num_frames: Optional[int]) -> torch.FloatTensor:
        """(B, T_a) → (B, num_frames, hidden or feature_dim)."""
        e = self._get_engine(1, num_frames or 1)
        return e.audio_encode(waveform, num_frames, apply_proj=self.feature_dim > 0)

    def _encode_distinct(self, waveform: torch.FloatTensor, num_frames: int) -> torch.FloatTensor:
        """get_audio_embedding with byte-identical rows encoded once.  Candidates are found from two cheap per-row checksums (two passes over
        the batch: a sort of whole 160,000-sample rows would cost more than it saves on a batch of distinct clips) and then compared exactly."""
        B = waveform.shape[0]
        ramp = torch.linspace(0.5, 1.5, waveform.shape[1], device=waveform.device, dtype=waveform.dtype)
        keys = torch.stack([waveform.sum(1), (waveform * ramp).sum(1)], 1)
        uniq, inverse = torch.unique(keys, dim=0, return_inverse=True)
        if uniq.shape[0] == B:
            return self.get_audio_embedding(waveform, num_frames)
        first = torch.full((uniq.shape[0],), B, dtype=torch.long, device=waveform.device).scatter_reduce_(
            0, inverse, torch.arange(B, device=waveform.device), reduce="amin")
        if not bool((waveform[first[inverse]] == waveform).all()):   # equal checksums, different samples: no shortcut
            return self.get_audio_embedding(waveform, num_frames)
        return self.get_audio_embedding(waveform[first].contiguous(), num_frames).index_select(0, inverse)

    def get_random_timesteps(self, batch_size: int) -> torch.LongTensor:
        return torch.randint(0, self.noise_scheduler.config.num_train_timesteps, (batch_size,), dtype=torch.long)

    def add_noise(self, sample: torch.FloatTensor, timestep: torch.LongTensor, noise: Optional[torch.Tensor] = None) -> SAIDNoiseAdditionOutput:
        self._get_engine(sample.shape[0], sample.shape[1])
        if noise is None:
            noise = torch.randn(sample.shape, device=sample.device)
        noisy_sample = self.noise_scheduler.add_noise(sample, noise, timestep)
        velocity = self.noise_scheduler.get_velocity(sample, noise, timestep)
        return SAIDNoiseAdditionOutput(noisy_sample=noisy_sample, noise=noise, velocity=velocity)

    def encode_samples(self, samples: torch.FloatTensor) -> torch.FloatTensor:
        return samples.clone()

    def decode_latent(self, latent: torch.FloatTensor) -> torch.FloatTensor:
        return latent.clone()

    def _pick_clip_groups(self, batch_size: int, tokens_per_clip: int) -> int:
        """How many concurrent clip groups SAID.inference runs a batch as (tokens_per_clip: UNet rows per clip, 2 T under
        guidance).  Clip groups are round 3's answer to launches that could not fill the chip on their own: contiguous sub-batches, each
        a complete denoising loop on its own stream.
        * bf16 mode, large batches (round 4): ONE group.  The persistent GEMMs (rgemm.hip: <= 256 workgroups for the whole batch, weights
          in registers, helper waves) keep every CU busy without a second chain beside them: 1.32 ms per step unsplit against 1.30 / 1.40
          with two / four groups at 32 clips x 600 frames (round 3's kernels: 1.99 unsplit, 1.73 with four) — and an unsplit batch does
          not depend on free hardware queues.
        * fp32 mode (measured on round 3's kernels, scripts/clip_groups_sweep.py): a split pays when every group keeps >= 12000 UNet rows,
          i.e. stays on the large-batch GEMM kernels with a full wave of workgroups: -4 .. -14 % per step; three groups beat two by 1-2 points.
        * small batches of 4+ clips that stay on the small-batch kernels as a whole are chains of short latency-bound launches: two such
          chains side by side overlap almost freely (-5 .. -14 % per step)."""
        if self.clip_groups is not None:
            return max(1, min(int(self.clip_groups), batch_size))
        bf = self.mfma_dtype == "bf16"
        if not bf:
            for g in (3, 2):
                if batch_size >= g and (batch_size // g) * tokens_per_clip >= 12000:
                    return g
        # At least two clips per group (single short clips side by side: +5 %); bf16 from 3000 rows on runs the persistent kernels: one group
        # (3 clips x 600 frames under guidance: 22.7 ms per 50 steps unsplit on them against 30.3 on the small-batch kernels).
        if batch_size >= 4 and batch_size * tokens_per_clip < (3000 if bf else 10000):
            return 2
        return 1

    def _group_engines(self, eng: "_engine.Engine", n: int, max_batch_eff: int, frames: int) -> List["_engine.Engine"]:
        frames_r = (frames + 63) // 64 * 64
        self._clones = [c for c in self._clones if c.h is not None]
        for c in self._clones[:n]:
            if c.max_batch_eff < max_batch_eff or c.max_frames < frames:
                c.reserve(max(max_batch_eff, c.max_batch_eff), max(frames_r, c.max_frames))
        while len(self._clones) < n:
            self._clones.append(eng.clone(max_batch_eff, frames_r))
        for c in self._clones[:n]:
            c.set_precision(self.mfma_dtype)
        return self._clones[:n]

    def inference(self, waveform_processed: torch.FloatTensor, init_samples: Optional[torch.FloatTensor] = None,
                  mask: Optional[torch.FloatTensor] = None, num_inference_steps: int = 100, strength: float = 1.0,
                  guidance_scale: float = 2.5, guidance_rescale: float = 0.0, eta: float = 0.0, fps: int = 60,
                  save_intermediate: bool = False, show_process: bool = False, *,
                  init_latents: Optional[torch.Tensor] = None, edit_noise: Optional[torch.Tensor] = None,
                  step_noise: Optional[torch.Tensor] = None,
                  audio_embedding: Optional[torch.Tensor] = None) -> SAIDInferenceOutput:
        """Inference pipeline (diffusion.py:308-472): schedule, start noise, optional init/mask
        editing, classifier-free guidance, DDIM updates, final clamp to [0, 1]."""
        batch_size, waveform_len = waveform_processed.shape
        in_channels = self.denoiser.in_channels
        device = waveform_processed.device
        do_cfg = guidance_scale > 1.0
        window_size = int(waveform_len / self.sampling_rate * fps)
        eng = self._get_engine(2 * batch_size if do_cfg else batch_size, window_size)
        sch = self.noise_scheduler
        sch.set_timesteps(num_inference_steps)

        if init_samples is None:
            latents = init_latents.to(device) if init_latents is not None else torch.randn(batch_size, window_size, in_channels, device=device)
        else:
            latents = self.encode_samples(init_samples)
        scale0 = self.latent_scale * sch.init_noise_sigma
        if scale0 != 1.0:
            latents = eng.axpby([scale0] * batch_size, latents)
        init_lat = latents.clone()
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)

        noise = None
        if init_samples is not None:
            t0 = sch.timesteps[-init_timestep]
            timesteps0 = torch.tensor([int(t0)] * batch_size, dtype=torch.long)
            noise = edit_noise.to(device) if edit_noise is not None else torch.randn(latents.shape, device=device)
            latents = sch.add_noise(latents, noise, timesteps0)

        if audio_embedding is None:
            # identical rows (the reference's batched caller repeats one clip 64 times, script/test_inference.py:167-168): encode each distinct
            # waveform once and gather — a clip's features do not depend on its batch neighbours
            audio_embedding = self._encode_distinct(waveform_processed, window_size) if (batch_size > 1 and self.dedupe_audio) \
                else self.get_audio_embedding(waveform_processed, window_size)

        t_start = num_inference_steps - init_timestep
        ts = sch.timesteps[t_start:].cpu().numpy().astype(np.int64)
        n_run = len(ts)
        coef = sch.coef_table(ts, float(eta))
        noise_steps, noise_seed = None, None
        if eta > 0 and n_run > 0:
            if step_noise is not None:   # injected by the caller (tests: the CPU oracle is fed the same draws)
                noise_steps = step_noise.to(device)
            else:
                # The reference draws randn(model_output.shape) inside scheduler.step, once per step (diffusion.py:441-443).
                # Here the step's last kernel generates its own standard normals from a counter-based generator keyed by
                # ONE 64-bit seed taken from torch's default generator (so torch.manual_seed still fixes the run): no
                # (N, B, T, 32) noise tensor exists — 2.46 GB at B = 32, N = 1000.
                noise_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
        use_mask = init_samples is not None and mask is not None
        mask_d = mask.to(device) if use_mask else None
        if mask_d is not None and mask_d.shape != latents.shape:
            mask_d = mask_d.expand_as(latents)
        kw = dict(timesteps=ts, coef=coef, prediction_type=sch.config.prediction_type, guidance_scale=guidance_scale,
                  guidance_rescale=guidance_rescale, latent_scale=self.latent_scale, save_intermediate=save_intermediate, noise_seed=noise_seed)

        def job(e, lo, hi):
            sl = slice(lo, hi)
            return e.loop_job(concurrent=(hi - lo) < batch_size, latents=latents[sl], context=audio_embedding[sl],
                              step_noise=None if noise_steps is None else noise_steps[:, sl].contiguous(),
                              init_latents=init_lat[sl] if use_mask else None, edit_noise=noise[sl] if use_mask else None,
                              mask=mask_d[sl] if use_mask else None, noise_batch_offset=lo, **kw)

        # Clip groups: a large batch runs as G (two or three) concurrent sub-batches, each on its own stream and engine context (the extra
        # contexts share the packed weights: said_clone).  One launch is a single wave of workgroups that load, multiply and
        # store in lockstep; concurrent loops interleave those phases: -7 % (bf16) / -5 % (fp32) per step at 32 clips x 600
        # frames on one MI355X.  The results are those of the sub-batches run alone (bit-identical to the whole batch in bf16
        # mode; fp32's GEMM tile depends on the launch size, so there sums differ in their last bits: 2e-5 after 6 steps).  The
        # eta noise is the whole batch's (noise_batch_offset).
        G = self._pick_clip_groups(batch_size, (2 if do_cfg else 1) * window_size)
        progress = _Progress(eng, n_run) if (show_process and n_run > 0) else None
        try:
            result, inter = self._run_groups(G, eng, job, batch_size, do_cfg, window_size, n_run, save_intermediate, device)
            if self.on_nonfinite != "ignore" and n_run > 0:
                # One synchronisation per call: did a step's model output overflow?  (The final clamp to [0, 1] would otherwise be all a caller sees.)
                used = [eng] + (self._clones[:G - 1] if G > 1 else [])
                bad = self._nonfinite(used)
                if bad is not None and eng.effective_precision() == "fp32":
                    if self.on_nonfinite == "raise":
                        raise _engine.EngineError(f"SAID.inference: {bad} in fp32 mode (split-fp16 products, operand domain |x| < 65504); use set_mfma_dtype('fp32_strict')")
                    warnings.warn(f"SAID.inference: {bad} on split-fp16 products (operand domain |x| < 65504); running the call again on fp32 matrix instructions "
                                  "with the same random draws (set_mfma_dtype('fp32_strict') avoids the first attempt)", RuntimeWarning, stacklevel=2)
                    old, self.mfma_dtype = self.mfma_dtype, "fp32_strict"
                    try:
                        eng.set_precision("fp32_strict")
                        result, inter = self._run_groups(G, eng, job, batch_size, do_cfg, window_size, n_run, save_intermediate, device)
                        bad = self._nonfinite(used)
                    finally:
                        self.mfma_dtype = old
                        eng.set_precision(old)
                if bad is not None:   # bf16 / strict fp32: the reference's own fp32 evaluation would overflow likewise (or bf16's range did)
                    warnings.warn(f"SAID.inference: {bad} ({eng.effective_precision()} mode)", RuntimeWarning, stacklevel=2)
            if progress is not None:
                torch.cuda.current_stream(device).synchronize()   # (the reference's loop is synchronous: the bar ends when the result exists)
        finally:
            if progress is not None:
                progress.close()
        intermediates = [inter[k] for k in range(n_run)] if save_intermediate else []
        return SAIDInferenceOutput(result=result, intermediates=intermediates)

    def _run_groups(self, G, eng, job, batch_size, do_cfg, window_size, n_run, save_intermediate, device):
        if G > 1 and n_run > 0:
            bounds = [batch_size * i // G for i in range(G + 1)]
            engines = [eng] + self._group_engines(eng, G - 1, (2 if do_cfg else 1) * max(bounds[i + 1] - bounds[i] for i in range(G)), window_size)
            main = torch.cuda.current_stream(device)
            streams = [main] + [e.stream for e in engines[1:]]       # each clone owns its stream
            # Outputs are allocated and the step graphs built here, in turn (stream capture and the allocator's hipMalloc must
            # not run beside another thread's launches); then every further group is ENQUEUED from its own host thread: a
            # 1000-step loop is 100 graph launches, more than a stream's queue takes without blocking the caller, so one thread
            # would enqueue (and the GPU run) the groups one after the other.  (ctypes releases the GIL inside said_denoise_loop.)
            jobs = []
            for i in range(G):
                if i:
                    streams[i].wait_stream(main)          # the inputs were produced on the caller's stream
                with torch.cuda.stream(streams[i]):
                    jobs.append(job(engines[i], bounds[i], bounds[i + 1]))
                    engines[i].prepare_loop(jobs[i])
            out, errs = [None] * G, []

            def enqueue(i):
                try:
                    with torch.cuda.device(device), torch.cuda.stream(streams[i]):
                        out[i] = engines[i].run_loop(jobs[i])
                except BaseException as ex:   # re-raised on the caller's thread
                    errs.append(ex)

            threads = [threading.Thread(target=enqueue, args=(i,), name=f"said-clip-group-{i}") for i in range(1, G)]
            for th in threads:
                th.start()
            try:
                enqueue(0)
            finally:
                for th in threads:
                    th.join()
            if errs:
                raise errs[0]
            for i in range(1, G):
                main.wait_stream(streams[i])
                for t in (out[i][0], out[i][2]):          # allocated on the side stream, consumed on the caller's
                    if t is not None:
                        t.record_stream(main)
            result = torch.cat([o[0] for o in out])
            inter = torch.cat([o[2] for o in out], dim=1) if save_intermediate else None
        else:
            result, _, inter = eng.run_loop(job(eng, 0, batch_size))
        return result, inter


class SAID_UNet1D(SAID):
    """SAiD model implemented using U-Net 1D model"""

    def __init__(self, audio_config=None, audio_processor=None, noise_scheduler: Type = DDIMScheduler, in_channels: int = 32,
                 feature_dim: int = -1, diffusion_steps: int = 1000, latent_scale: float = 1,
                 prediction_type: str = "epsilon"):
        # NB: like the reference (diffusion.py:510-518) `noise_scheduler` is accepted but not
        # forwarded, so the scheduler is always DDIM.
        super().__init__(audio_config=audio_config, audio_processor=audio_processor, in_channels=in_channels,
                         feature_dim=feature_dim, diffusion_steps=diffusion_steps, latent_scale=latent_scale,
                         prediction_type=prediction_type)
        self.denoiser = UNet1DConditionModel(
            in_channels=in_channels, out_channels=in_channels,
            cross_attention_dim=self.feature_dim if self.feature_dim > 0 else self.audio_config.hidden_size)
        self.denoiser._owner = weakref.ref(self)
