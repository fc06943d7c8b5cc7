"""ctypes binding of libsaid_hip.so (C ABI: include/said_hip.h).

There is deliberately no fallback: if the library is missing, cannot be loaded,
or no gfx950 device is visible, every entry point raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_void_p
from typing import Dict, Optional, Sequence

import numpy as np
import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsaid_hip.so")
_lib = None

PRED = {"epsilon": 0, "sample": 1, "v_prediction": 2}
PRECISIONS = {"fp32": 0, "bf16": 1, "fp32_strict": 2}   # SAID_PREC_* of include/said_hip.h
NCOEF = 8
ABI_VERSION = 9   # include/said_hip.h as bound below; a stale libsaid_hip.so is refused at load time


class EngineError(RuntimeError):
    pass


class LoopParams(ctypes.Structure):
    _fields_ = [
        ("batch", c_int), ("frames", c_int), ("num_steps", c_int), ("prediction_type", c_int),
        ("guidance_scale", c_float), ("guidance_rescale", c_float), ("latent_scale", c_float),
        ("use_step_noise", c_int), ("use_mask", c_int), ("save_intermediate", c_int),
        ("timesteps_host", POINTER(c_int64)), ("coef_host", POINTER(c_float)),
        ("context_dev", c_void_p), ("latents_dev", c_void_p), ("step_noise_dev", c_void_p),
        ("init_latents_dev", c_void_p), ("edit_noise_dev", c_void_p), ("mask_dev", c_void_p),
        ("intermediates_dev", c_void_p), ("result_dev", c_void_p), ("noise_seed", ctypes.c_uint64),
        ("noise_batch_offset", c_int), ("concurrent", c_int),
    ]


EXPORTS = {
    "said_abi_version": (c_int, []),
    "said_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int]),
    "said_destroy": (c_int, [c_void_p]),
    "said_reserve": (c_int, [c_void_p, c_int, c_int]),
    "said_capacity": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "said_clone": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_int]),
    "said_stream": (c_void_p, [c_void_p]),
    "said_loop_prepare": (c_int, [c_void_p, POINTER(LoopParams), c_void_p]),
    "said_last_error": (c_char_p, [c_void_p]),
    "said_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "said_finalize_weights": (c_int, [c_void_p, c_void_p]),
    "said_set_timestep_freqs": (c_int, [c_void_p, c_void_p, c_int]),
    "said_audio_encode": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, POINTER(c_int), c_void_p]),
    "said_unet_forward": (c_int, [c_void_p, c_void_p, POINTER(c_int64), c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "said_denoise_loop": (c_int, [c_void_p, POINTER(LoopParams), c_void_p]),
    "said_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, POINTER(c_float), c_int, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "said_axpby": (c_int, [c_void_p, POINTER(c_float), c_void_p, POINTER(c_float), c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "said_graph_num_nodes": (c_int, [c_void_p]),
    "said_loop_progress": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "said_loop_progress_reset": (c_int, [c_void_p]),
    "said_set_precision": (c_int, [c_void_p, c_int]),
    "said_get_precision": (c_int, [c_void_p]),
    "said_effective_precision": (c_int, [c_void_p]),
    "said_precision_note": (c_char_p, [c_void_p]),
    "said_numeric_status": (c_int, [c_void_p, c_void_p, POINTER(c_int), POINTER(c_int)]),
    "said_profile_unet": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, POINTER(c_int), c_void_p]),
    "said_philox_normal": (c_int, [c_void_p, ctypes.c_uint64, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "said_debug_option": (c_int, [c_void_p, c_char_p, ctypes.c_longlong]),
    "said_debug_get": (ctypes.c_longlong, [c_void_p, c_char_p]),
    "said_debug_stop_after": (c_int, [c_void_p, c_int]),
    "said_debug_clocks": (c_int, [c_void_p, c_int, c_void_p]),
    "said_debug_read": (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    "said_debug_ws_count": (c_int, [c_void_p]),
    "said_debug_ws_info": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(ctypes.c_longlong), POINTER(c_char_p)]),
    "said_debug_ws_fill": (c_int, [c_void_p, c_int]),
    "said_debug_ws_copy": (c_int, [c_void_p, c_int, c_void_p, ctypes.c_longlong, c_void_p]),
    "said_unet_algorithmic_bytes": (c_double, [c_int, c_int, c_int]),
    "said_unet_algorithmic_flops": (c_double, [c_int, c_int]),
    "said_vae_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int]),
    "said_vae_destroy": (c_int, [c_void_p]),
    "said_vae_last_error": (c_char_p, [c_void_p]),
    "said_vae_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "said_vae_finalize_weights": (c_int, [c_void_p]),
    "said_vae_encode": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p, c_void_p, c_void_p]),
}


def library_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen the engine and bind every symbol of include/said_hip.h (no compute)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise EngineError(
            f"{_LIB_PATH} not found: build it with `python -m said_amd.build` (hipcc, gfx950). "
            "said_amd has no CPU fallback.")
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.said_abi_version()
    if got != ABI_VERSION:
        raise EngineError(f"{_LIB_PATH} has ABI version {got}, this binding expects {ABI_VERSION}: rebuild it with "
                          "`python -m said_amd.build --force`")
    _lib = lib
    return lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _check_dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise EngineError(f"{name} must live on the MI355X (got device {t.device}); said_amd has no CPU path")
    if t.dtype != torch.float32:
        raise EngineError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    """One engine context on one GPU (one process per GPU)."""

    def __init__(self, device: torch.device, max_batch_eff: int, max_frames: int, in_channels: int = 32, ctx_dim: int = 768, _clone_of=None):
        self.lib = load_library()
        device = torch.device(device)
        if device.type != "cuda":
            raise EngineError(f"said_amd runs on MI355X only (device={device}); there is no CPU fallback")
        self.device = device
        self.index = device.index if device.index is not None else torch.cuda.current_device()
        self.max_batch_eff, self.max_frames = int(max_batch_eff), int(max_frames)
        self.in_channels, self.ctx_dim = in_channels, ctx_dim
        h = c_void_p()
        self._parent = _clone_of   # a clone shares its parent's packed weights: keep the parent alive, destroy the clone first
        if _clone_of is not None:
            rc = self.lib.said_clone(_clone_of.h, ctypes.byref(h), self.max_batch_eff, self.max_frames)
            if rc != 0:
                raise EngineError("said_clone: " + (self.lib.said_last_error(_clone_of.h) or b"?").decode())
        else:
            rc = self.lib.said_create(ctypes.byref(h), self.index, self.max_batch_eff, self.max_frames, in_channels, ctx_dim)
            if rc != 0:
                raise EngineError("said_create: " + (self.lib.said_last_error(None) or b"?").decode())
        self.h = h
        self.has_audio = _clone_of.has_audio if _clone_of is not None else False
        self._clones = []
        self._keep = []  # host buffers referenced by in-flight async copies
        sp = self.lib.said_stream(self.h) if _clone_of is not None else None
        self.stream = torch.cuda.ExternalStream(sp, device=self.device) if sp else None   # a clone's own stream

    def clone(self, max_batch_eff: int, max_frames: int) -> "Engine":
        """A context sharing this one's packed weights, with its own workspace and step graph (said_clone): for concurrent
        denoising loops on a second stream.  Closed together with this engine."""
        c = Engine(self.device, max_batch_eff, max_frames, self.in_channels, self.ctx_dim, _clone_of=self)
        self._clones.append(c)
        return c

    def close(self):
        for c in getattr(self, "_clones", []):
            c.close()      # clones first: they point into this context's weights
        self._clones = []
        if getattr(self, "h", None):
            self.lib.said_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int, what: str):
        if rc != 0:
            raise EngineError(f"{what}: " + (self.lib.said_last_error(self.h) or b"?").decode())

    def reserve(self, max_batch_eff: int, max_frames: int) -> None:
        """Grow the workspace (never shrinks); the packed weights stay on the device (said_reserve)."""
        with torch.cuda.device(self.index):
            rc = self.lib.said_reserve(self.h, int(max_batch_eff), int(max_frames))
        msg = (self.lib.said_last_error(self.h) or b"?").decode() if rc != 0 else ""
        b, t = c_int(0), c_int(0)
        self._chk(self.lib.said_capacity(self.h, ctypes.byref(b), ctypes.byref(t)), "said_capacity")
        self.max_batch_eff, self.max_frames = b.value, t.value   # (0, 0) after a failed growth: the context refuses every size
        if rc != 0:
            raise EngineError("said_reserve: " + msg)

    # ---- weights ----
    def load_weights(self, state_dict: Dict[str, torch.Tensor]):
        """`state_dict` uses the reference's SAID key layout."""
        with torch.cuda.device(self.index):
            half = 96
            freqs = torch.exp(-np.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
            fr = np.ascontiguousarray(freqs.numpy())
            self._chk(self.lib.said_set_timestep_freqs(self.h, fr.ctypes.data_as(c_void_p), half), "said_set_timestep_freqs")
            for k, v in state_dict.items():
                a = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy())
                shape = (c_int64 * a.ndim)(*a.shape)
                self._chk(self.lib.said_set_weight(self.h, k.encode(), a.ctypes.data_as(c_void_p), shape, a.ndim), f"said_set_weight({k})")
            self._chk(self.lib.said_finalize_weights(self.h, _stream()), "said_finalize_weights")
        self.has_audio = any(k.startswith("audio_encoder.") for k in state_dict)

    # ---- compute ----
    def audio_encode(self, waveform: torch.Tensor, num_frames: Optional[int], apply_proj: bool = False) -> torch.Tensor:
        waveform = _check_dev(waveform, "waveform")
        B, Ta = waveform.shape
        out_dim = self.ctx_dim if apply_proj else 768
        with torch.cuda.device(self.index):
            # frame count without interpolation is decided by the library
            nf = int(num_frames) if num_frames is not None else 0
            if nf > 0:
                out = torch.empty(B, nf, out_dim, device=waveform.device, dtype=torch.float32)
            else:
                L = Ta
                for k, s in zip((10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)):
                    L = (L - k) // s + 1
                out = torch.empty(B, max(L, 1), out_dim, device=waveform.device, dtype=torch.float32)
            got = c_int(0)
            self._chk(self.lib.said_audio_encode(self.h, _ptr(waveform), B, Ta, nf, int(apply_proj), _ptr(out), ctypes.byref(got), _stream()), "said_audio_encode")
            assert got.value == out.shape[1], (got.value, out.shape)
        return out

    def unet_forward(self, sample: torch.Tensor, timesteps: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        sample = _check_dev(sample, "sample")
        context = _check_dev(context, "encoder_hidden_states")
        Be, T, C = sample.shape
        if context.shape[0] != Be or context.shape[2] != self.ctx_dim:
            raise EngineError(f"context shape {tuple(context.shape)} does not match batch {Be} / ctx_dim {self.ctx_dim}")
        ts = np.ascontiguousarray(timesteps.detach().to("cpu", torch.int64).reshape(-1).numpy())
        if ts.shape[0] != Be:
            raise EngineError(f"timesteps must have {Be} entries, got {ts.shape[0]}")
        out = torch.empty_like(sample)
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_unet_forward(self.h, _ptr(sample), ts.ctypes.data_as(POINTER(c_int64)), _ptr(context), Be, T,
                                                 context.shape[1], _ptr(out), _stream()), "said_unet_forward")
        return out

    def loop_job(self, *, latents: torch.Tensor, context: torch.Tensor, timesteps: np.ndarray, coef: np.ndarray,
                     prediction_type: str, guidance_scale: float, guidance_rescale: float, latent_scale: float,
                     step_noise: Optional[torch.Tensor] = None, init_latents: Optional[torch.Tensor] = None,
                     edit_noise: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                     save_intermediate: bool = False, noise_seed: Optional[int] = None, noise_batch_offset: int = 0, concurrent: bool = False):
        """Validates the arguments and allocates the outputs of one denoising loop (on the current stream): a job for
        prepare_loop / run_loop.  `noise_seed` (with step_noise None): the eta noise is generated inside the step's last
        kernel (Philox4x32-10 keyed by the seed) instead of being read from a tensor."""
        latents = _check_dev(latents, "latents").clone()
        context = _check_dev(context, "audio_embedding")
        B, T, C = latents.shape
        N = int(len(timesteps))
        ts = np.ascontiguousarray(np.asarray(timesteps, dtype=np.int64))
        cf = np.ascontiguousarray(np.asarray(coef, dtype=np.float32).reshape(N, NCOEF))
        result = torch.empty_like(latents)
        inter = torch.empty(N, B, T, C, device=latents.device, dtype=torch.float32) if save_intermediate else None
        use_mask = mask is not None and init_latents is not None
        keep = [ts, cf, latents, context, result, inter]
        p = LoopParams()
        p.batch, p.frames, p.num_steps, p.prediction_type = B, T, N, PRED[prediction_type]
        p.guidance_scale, p.guidance_rescale, p.latent_scale = float(guidance_scale), float(guidance_rescale), float(latent_scale)
        p.use_step_noise = int(step_noise is not None)
        p.use_mask = int(use_mask)
        p.save_intermediate = int(save_intermediate)
        p.concurrent = int(bool(concurrent))
        p.timesteps_host = ts.ctypes.data_as(POINTER(c_int64))
        p.coef_host = cf.ctypes.data_as(POINTER(c_float))
        p.context_dev = context.data_ptr()
        p.latents_dev = latents.data_ptr()
        if step_noise is not None:
            step_noise = _check_dev(step_noise, "step_noise")
            assert tuple(step_noise.shape) == (N, B, T, C), step_noise.shape
            p.step_noise_dev = step_noise.data_ptr()
            keep.append(step_noise)
        elif noise_seed is not None:
            p.use_step_noise = 2
            p.noise_seed = int(noise_seed) & 0xFFFFFFFFFFFFFFFF
            p.noise_batch_offset = int(noise_batch_offset)
        if use_mask:
            init_latents = _check_dev(init_latents, "init_latents")
            edit_noise = _check_dev(edit_noise, "edit_noise")
            mask = _check_dev(mask.expand_as(latents) if mask.shape != latents.shape else mask, "mask")
            p.init_latents_dev, p.edit_noise_dev, p.mask_dev = init_latents.data_ptr(), edit_noise.data_ptr(), mask.data_ptr()
            keep += [init_latents, edit_noise, mask]
        if inter is not None:
            p.intermediates_dev = inter.data_ptr()
        p.result_dev = result.data_ptr()
        return p, keep, (result, latents, inter)

    def prepare_loop(self, job):
        """Builds the job's step graph if this context does not hold it yet (said_loop_prepare); launches nothing of the loop."""
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_loop_prepare(self.h, ctypes.byref(job[0]), _stream()), "said_loop_prepare")

    def run_loop(self, job):
        """Enqueues the job's loop on the current stream; returns (result, final_latents, intermediates or None)."""
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_denoise_loop(self.h, ctypes.byref(job[0]), _stream()), "said_denoise_loop")
        self._keep = job[1]  # alive until the next call (async copies / kernels may still reference them)
        return job[2]

    def denoise_loop(self, **kw):
        """loop_job + run_loop: returns (result, final_latents, intermediates or None)."""
        return self.run_loop(self.loop_job(**kw))

    def ddim_step(self, eps: torch.Tensor, sample: torch.Tensor, coef_row: np.ndarray, prediction_type: str,
                  eps_uncond: Optional[torch.Tensor] = None, guidance_scale: float = 1.0,
                  step_noise: Optional[torch.Tensor] = None, init_latents: Optional[torch.Tensor] = None,
                  edit_noise: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        eps, sample = _check_dev(eps, "model_output"), _check_dev(sample, "sample")
        out = torch.empty_like(sample)
        cf = np.ascontiguousarray(np.asarray(coef_row, dtype=np.float32).reshape(NCOEF))
        opt = [None if t is None else _check_dev(t, "tensor") for t in (eps_uncond, step_noise, init_latents, edit_noise, mask)]
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_ddim_step(self.h, _ptr(eps), _ptr(opt[0]), float(guidance_scale), _ptr(sample),
                                              cf.ctypes.data_as(POINTER(c_float)), PRED[prediction_type], _ptr(opt[1]), _ptr(opt[2]),
                                              _ptr(opt[3]), _ptr(opt[4]), _ptr(out), sample.numel(), _stream()), "said_ddim_step")
            torch.cuda.current_stream().synchronize()  # cf is a temporary
        return out

    def axpby(self, a: Sequence[float], x: torch.Tensor, c: Optional[Sequence[float]] = None, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = _check_dev(x, "x")
        B = x.shape[0]
        if len(a) != B or (c is not None and len(c) != B):
            raise EngineError(f"axpby: need one coefficient per sample ({B}), got {len(a)}" + (f" / {len(c)}" if c is not None else ""))
        av = (c_float * B)(*[float(v) for v in a])
        cv = (c_float * B)(*[float(v) for v in (c if c is not None else [0.0] * B)])
        if y is not None:
            y = _check_dev(y, "y")
        out = torch.empty_like(x)
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_axpby(self.h, av, _ptr(x), cv, _ptr(y), _ptr(out), B, x.numel() // B, _stream()), "said_axpby")
        return out

    def profile_unet(self, batch_eff: int, frames: int, reps: int = 50, cfg_clips: int = 0):
        """Per-launch (us, bytes, flops, kind, epi, NB, KS) of the UNet kernel schedule, HIP-event timed."""
        M = 128
        us = np.zeros(M, np.float32); by = np.zeros(M, np.float64); fl = np.zeros(M, np.float64)
        kind = np.zeros(M, np.int32); epi = np.zeros(M, np.int32); nb = np.zeros(M, np.int32); ks = np.zeros(M, np.int32)
        n = c_int(0)
        vp = lambda a: a.ctypes.data_as(c_void_p)
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_profile_unet(self.h, batch_eff, frames, cfg_clips, reps, M, vp(us), vp(by), vp(fl), vp(kind), vp(epi), vp(nb),
                                                 vp(ks), ctypes.byref(n), _stream()), "said_profile_unet")
        k = n.value
        return [dict(us=float(us[i]), bytes=float(by[i]), flops=float(fl[i]), kind=int(kind[i]), epi=int(epi[i]), NB=int(nb[i]), KS=int(ks[i]))
                for i in range(k)]

    def philox_normal(self, seed: int, step0: int, nsteps: int, shape) -> torch.Tensor:
        """(nsteps, *shape) standard normals: exactly what denoise_loop(noise_seed=seed) adds at steps step0 .. step0 + nsteps - 1."""
        n = int(np.prod(shape))
        out = torch.empty((nsteps,) + tuple(shape), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_philox_normal(self.h, int(seed) & 0xFFFFFFFFFFFFFFFF, int(step0), int(nsteps), n, _ptr(out), _stream()),
                      "said_philox_normal")
        return out

    # ---- debugging aids (tests only) ----
    def debug_option(self, name: str, value: int) -> None:
        self._chk(self.lib.said_debug_option(self.h, name.encode(), int(value)), "said_debug_option")

    def debug_get(self, name: str) -> int:
        return int(self.lib.said_debug_get(self.h, name.encode()))

    def debug_stop_after(self, n: int):
        self._chk(self.lib.said_debug_stop_after(self.h, int(n)), "said_debug_stop_after")

    def debug_clocks(self, enable: bool, read: bool = False):
        out = np.zeros((64, 8, 16), dtype=np.int64) if read else None
        self._chk(self.lib.said_debug_clocks(self.h, int(enable), out.ctypes.data_as(c_void_p) if read else None), "said_debug_clocks")
        return out

    def debug_read(self, name: str, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float32)
        self._chk(self.lib.said_debug_read(self.h, name.encode(), out.ctypes.data_as(c_void_p), out.size), "said_debug_read")
        return out

    def ws_buffers(self):
        """[(index, name, bytes)] of the context's workspace buffers (said_debug_ws_info)."""
        out = []
        for i in range(int(self.lib.said_debug_ws_count(self.h))):
            p, nb, nm = c_void_p(), ctypes.c_longlong(0), c_char_p()
            self._chk(self.lib.said_debug_ws_info(self.h, i, ctypes.byref(p), ctypes.byref(nb), ctypes.byref(nm)), "said_debug_ws_info")
            out.append((i, (nm.value or b"?").decode(), int(nb.value)))
        return out

    def ws_fill(self, byte_value: int) -> None:
        self._chk(self.lib.said_debug_ws_fill(self.h, int(byte_value) & 0xFF), "said_debug_ws_fill")

    def ws_snapshot(self, idx: int, nbytes: int) -> torch.Tensor:
        """Device copy (uint8) of workspace buffer `idx`, enqueued on the current stream."""
        out = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_debug_ws_copy(self.h, idx, _ptr(out), nbytes, _stream()), "said_debug_ws_copy")
        return out

    def loop_progress(self) -> int:
        """Denoise steps started so far by the loop running (or last run) on this context; never blocks the loop's stream."""
        v = c_int(0)
        if self.lib.said_loop_progress(self.h, ctypes.byref(v)) != 0:   # (the entry does not record an error text: it runs beside the loop's thread)
            raise EngineError("said_loop_progress failed")
        return v.value

    def loop_progress_reset(self) -> None:
        """Forget the previous loop's step count before a polling thread is started."""
        self._chk(self.lib.said_loop_progress_reset(self.h), "said_loop_progress_reset")

    def graph_num_nodes(self) -> int:
        return int(self.lib.said_graph_num_nodes(self.h))

    def set_precision(self, mode) -> None:
        """"fp32" (split-fp16 products), "fp32_strict" (fp32 matrix instructions) or "bf16"; see said_set_precision.  (True / False: "bf16" / "fp32".)"""
        if isinstance(mode, str):
            if mode not in PRECISIONS:
                raise EngineError(f"unknown precision mode {mode!r}: one of {sorted(PRECISIONS)}")
            mode = PRECISIONS[mode]
        else:
            mode = 1 if mode else 0
        self._chk(self.lib.said_set_precision(self.h, int(mode)), "said_set_precision")

    def get_precision(self) -> str:
        """The mode asked for."""
        return {v: k for k, v in PRECISIONS.items()}[int(self.lib.said_get_precision(self.h))]

    def effective_precision(self) -> str:
        """The mode that runs: "fp32" becomes "fp32_strict" when a weight tensor lies outside the split-fp16 range (precision_note() says which)."""
        return {v: k for k, v in PRECISIONS.items()}[int(self.lib.said_effective_precision(self.h))]

    def precision_note(self) -> str:
        return (self.lib.said_precision_note(self.h) or b"").decode()

    def numeric_status(self):
        """(first_bad_step, result_nonfinite) of the last loop / forward call on this context: the index of the first denoise step whose model output
        held an inf / NaN (-1: none) and whether the final latents / the model output hold one.  Synchronises the current stream."""
        a, b = c_int(-1), c_int(0)
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_numeric_status(self.h, _stream(), ctypes.byref(a), ctypes.byref(b)), "said_numeric_status")
        return a.value, bool(b.value)


class VaeEngine:
    """BCVAE encoder context on one GPU (include/said_hip.h, "VAE encoder")."""

    def __init__(self, device: torch.device, in_channels: int = 32, seq_len: int = 120, z_dim: int = 64):
        self.lib = load_library()
        device = torch.device(device)
        if device.type != "cuda":
            raise EngineError(f"said_amd runs on MI355X only (device={device}); there is no CPU fallback")
        self.device = device
        self.index = device.index if device.index is not None else torch.cuda.current_device()
        self.seq_len, self.in_channels, self.z_dim = seq_len, in_channels, z_dim
        h = c_void_p()
        if self.lib.said_vae_create(ctypes.byref(h), self.index, in_channels, seq_len, z_dim) != 0:
            raise EngineError("said_vae_create: " + (self.lib.said_vae_last_error(None) or b"?").decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.said_vae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int, what: str):
        if rc != 0:
            raise EngineError(f"{what}: " + (self.lib.said_vae_last_error(self.h) or b"?").decode())

    def load_weights(self, state_dict: Dict[str, torch.Tensor]):
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue   # a counter, not a weight
            a = np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy())
            shape = (c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
            self._chk(self.lib.said_vae_set_weight(self.h, k.encode(), a.ctypes.data_as(c_void_p), shape, max(a.ndim, 1)), f"said_vae_set_weight({k})")
        self._chk(self.lib.said_vae_finalize_weights(self.h), "said_vae_finalize_weights")

    def encode(self, coeffs: torch.Tensor, n_windows: int, window_stride: int, want_logvar: bool = True):
        """`coeffs`: contiguous fp32 device tensor holding the windows at `window_stride` floats apart."""
        coeffs = _check_dev(coeffs, "coeffs")
        if coeffs.device.index != self.index:
            raise EngineError(f"coeffs live on cuda:{coeffs.device.index}, this VAE engine on cuda:{self.index}")
        need = (n_windows - 1) * window_stride + self.seq_len * self.in_channels if n_windows > 0 else 0
        if coeffs.numel() < need:
            raise EngineError(f"coeffs holds {coeffs.numel()} floats, {n_windows} windows at stride {window_stride} need {need}")
        mean = torch.empty(n_windows, self.z_dim, device=coeffs.device, dtype=torch.float32)
        logvar = torch.empty_like(mean) if want_logvar else None
        with torch.cuda.device(self.index):
            self._chk(self.lib.said_vae_encode(self.h, _ptr(coeffs), int(window_stride), int(n_windows), _ptr(mean), _ptr(logvar), _stream()),
                      "said_vae_encode")
        return mean, logvar


def unet_algorithmic_bytes(batch_eff: int, frames: int, bytes_per_elem: int = 4) -> float:
    return float(load_library().said_unet_algorithmic_bytes(batch_eff, frames, bytes_per_elem))


def unet_algorithmic_flops(batch_eff: int, frames: int) -> float:
    return float(load_library().said_unet_algorithmic_flops(batch_eff, frames))
