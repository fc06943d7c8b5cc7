"""BCVAE — encoder half on the MI355X engine.

Drop-in for the slice of /root/reference/said/model/vae.py that the evaluation driver uses
(script/test_evaluate.py:53-106, 551-554): ``BCVAE()`` with the reference's ``state_dict()`` key layout (70 tensors,
``encoder.*`` and ``decoder.*``, so ``load_state_dict(torch.load("vae.pth"))`` works strictly), ``.seq_len``,
``.eval()`` and ``.encode(coeffs) -> BCLatent(mean, log_var)``.  All encoder math runs in HIP (``said_vae_encode``);
modules only hold parameters.  The decoder / reparametrisation are training-time components and are out of scope
(SURVEY.md §2 row 8): their parameters are kept so that checkpoints load, calling them raises.

Extension (not in the reference): ``encode_windows`` runs all sliding windows of a sequence in ONE engine call instead
of the reference's Python loop over windows (test_evaluate.py:92-95).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from .. import _engine


@dataclass
class BCLatent:
    """Latent variables of the BCVAE"""

    mean: torch.FloatTensor
    log_var: torch.FloatTensor


class _ConvP(nn.Module):
    def __init__(self, cout: int, cin: int, k: int, transpose: bool = False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, k) if transpose else torch.empty(cout, cin, k))
        self.bias = nn.Parameter(torch.empty(cout))


class _BNP(nn.Module):
    def __init__(self, n: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _LinP(nn.Module):
    def __init__(self, cout: int, cin: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout))


class _Empty(nn.Module):
    pass


def _seq(entries):
    """nn.Sequential-like container with the reference's numeric child names; parameter-free slots stay empty."""
    m = nn.Module()
    for idx, child in entries:
        m.add_module(str(idx), child)
    return m


class BCEncoder(nn.Module):
    """Parameter container of vae.py:26-64 (Conv1d/BatchNorm1d/LeakyReLU x3, Conv1d, Flatten; Linear/BatchNorm1d/
    LeakyReLU x2, Linear; fc_mu, fc_logvar)."""

    def __init__(self, in_channels: int = 32, z_dim: int = 64):
        super().__init__()
        self.conv_layers = _seq([(0, _ConvP(32, in_channels, 3)), (1, _BNP(32)), (3, _ConvP(64, 32, 3)), (4, _BNP(64)),
                                 (6, _ConvP(64, 64, 4)), (7, _BNP(64)), (9, _ConvP(32, 64, 3))])
        self.fc_layers = _seq([(0, _LinP(256, 1760)), (1, _BNP(256)), (3, _LinP(128, 256)), (4, _BNP(128)), (6, _LinP(z_dim, 128))])
        self.fc_mu = _LinP(z_dim, z_dim)
        self.fc_logvar = _LinP(z_dim, z_dim)


class BCDecoder(nn.Module):
    """Parameter container of vae.py:115-178 (kept so that vae.pth loads strictly; not computed on this path)."""

    def __init__(self, out_channels: int = 32, seq_len: int = 120, z_dim: int = 64):
        super().__init__()
        self.fc_layers = _seq([(0, _LinP(2 * seq_len, z_dim)), (1, _BNP(2 * seq_len)), (3, _LinP(4 * seq_len, 2 * seq_len))])
        self.conv_layers = _seq([(0, _ConvP(32, 4, 3, transpose=True)), (1, _BNP(32)), (3, _ConvP(32, 32, 3, transpose=True)), (4, _BNP(32)),
                                 (6, _ConvP(32, 32, 3)), (7, _ConvP(out_channels, 32, 3))])


class BCVAE(nn.Module):
    """Autoencoder for the blendshape coefficients — encoder on the HIP engine (vae.py:181-272)."""

    def __init__(self, channels: int = 32, seq_len: int = 120, z_dim: int = 64):
        super().__init__()
        self.seq_len, self.channels, self.z_dim = seq_len, channels, z_dim
        self.encoder = BCEncoder(channels, z_dim)
        self.decoder = BCDecoder(channels, seq_len, z_dim)
        for n_, p in self.named_parameters():   # torch.empty holders: give them finite values until a checkpoint is loaded
            if p.dim() > 1:
                nn.init.normal_(p, std=0.02)
            elif n_.endswith("bias"):
                nn.init.zeros_(p)
        self._eng: Optional[_engine.VaeEngine] = None
        self._eng_key = None
        self._eng_stale = True

    # The engine holds folded copies of the weights (BatchNorm folded into the preceding layer).  load_state_dict() and
    # .to() / .cuda() (nn.Module._apply) mark them stale, as does an in-place update that bumps a tensor's version counter;
    # writes through `.data` bypass both: call refresh_engine() after them.
    def _apply(self, fn, *args, **kwargs):
        self._eng_stale = True
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._eng_stale = True
        return super().load_state_dict(*args, **kwargs)

    def refresh_engine(self) -> "BCVAE":
        self._eng_stale = True
        return self

    def _weights_key(self):
        ts = list(self.parameters()) + list(self.buffers())
        return (str(ts[0].device), tuple(t._version for t in ts))

    def _get_engine(self) -> _engine.VaeEngine:
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _engine.EngineError(f"model is on {dev}: said_amd runs on MI355X only — call .to('cuda:N')")
        if self.training:
            raise _engine.EngineError("BCVAE.encode on the HIP engine is eval-mode only (BatchNorm uses running statistics): call .eval()")
        key = self._weights_key()
        if self._eng is None or self._eng_stale or key != self._eng_key:
            if self._eng is not None:
                self._eng.close()
            e = _engine.VaeEngine(dev, self.channels, self.seq_len, self.z_dim)
            e.load_weights({k: v for k, v in self.state_dict().items() if k.startswith("encoder.")})
            self._eng, self._eng_key, self._eng_stale = e, key, False
        return self._eng

    def encode(self, coeffs: torch.Tensor) -> BCLatent:
        """(Batch_size, seq_len=120, 32) -> BCLatent(mean (B, 64), log_var (B, 64))   (vae.py:228-243)."""
        if coeffs.dim() != 3 or coeffs.shape[1] != self.seq_len or coeffs.shape[2] != self.channels:
            raise ValueError(f"coeffs must be (B, {self.seq_len}, {self.channels}), got {tuple(coeffs.shape)}")
        mean, logvar = self._get_engine().encode(coeffs.contiguous(), coeffs.shape[0], self.seq_len * self.channels)
        return BCLatent(mean=mean, log_var=logvar)

    def encode_windows(self, coeffs_seq: torch.Tensor, window_step_size: int, padding: int = 0) -> torch.Tensor:
        """Latent means of all sliding windows of one (T, 32) sequence, (num_windows, 64): the window loop of
        generate_latents_info (test_evaluate.py:89-95: num_windows = (T - seq_len) // step + 1 - padding) as one call."""
        if coeffs_seq.dim() == 3 and coeffs_seq.shape[0] == 1:
            coeffs_seq = coeffs_seq[0]
        T = coeffs_seq.shape[0]
        n = (T - self.seq_len) // window_step_size + 1 - padding
        if n <= 0:
            return torch.empty(0, self.z_dim, device=coeffs_seq.device)
        mean, _ = self._get_engine().encode(coeffs_seq.contiguous(), n, window_step_size * self.channels, want_logvar=False)
        return mean

    def forward(self, coeffs: torch.Tensor, use_noise: bool = True):
        raise NotImplementedError("BCVAE.forward (reparametrise + decode, vae.py:209-226) is a training-time path: out of scope "
                                  "(SURVEY.md §2 row 8); use .encode()")

    def decode(self, latent: torch.Tensor):
        raise NotImplementedError("BCVAE.decode (vae.py:258-272) is not on the evaluation path: out of scope (SURVEY.md §2 row 8)")

    def reparametrize(self, mean, log_var):
        raise NotImplementedError("BCVAE.reparametrize (vae.py:244-256) is a training-time path: out of scope")
