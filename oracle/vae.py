"""CPU restatement of the BCVAE encoder — TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline may import it).

``BCEncoder.forward`` (/root/reference/said/model/vae.py:26-83) in eval mode, as plain functional PyTorch on a state
dict with the reference's key names, and the sliding-window driver of ``generate_latents_info``
(/root/reference/script/test_evaluate.py:53-106).  Pinned to the reference's own ``BCVAE`` class by golden G10
(tests/golden/make_golden.py), with deterministic weights and — in the build container — with the reference's
``model/vae.pth``.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
SEQ_LEN = 120


def _bn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """nn.BatchNorm1d in eval mode: running statistics, eps 1e-5 (vae.py:42, 45, 48, 56, 59)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def encode(sd: SD, coeffs: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(B, 120, 32) -> (mean (B, 64), log_var (B, 64))   vae.py:66-83."""
    e = "encoder."
    x = coeffs.transpose(1, 2)                                                                        # :79
    x = F.leaky_relu(_bn(sd, e + "conv_layers.1", F.conv1d(x, sd[e + "conv_layers.0.weight"], sd[e + "conv_layers.0.bias"])), 0.2)   # :41-43
    x = F.leaky_relu(_bn(sd, e + "conv_layers.4", F.conv1d(x, sd[e + "conv_layers.3.weight"], sd[e + "conv_layers.3.bias"])), 0.2)   # :44-46
    x = F.leaky_relu(_bn(sd, e + "conv_layers.7", F.conv1d(x, sd[e + "conv_layers.6.weight"], sd[e + "conv_layers.6.bias"], stride=2)), 0.2)  # :47-49
    x = F.conv1d(x, sd[e + "conv_layers.9.weight"], sd[e + "conv_layers.9.bias"]).flatten(1)          # :50-51
    x = F.leaky_relu(_bn(sd, e + "fc_layers.1", F.linear(x, sd[e + "fc_layers.0.weight"], sd[e + "fc_layers.0.bias"])), 0.01)   # :55-57
    x = F.leaky_relu(_bn(sd, e + "fc_layers.4", F.linear(x, sd[e + "fc_layers.3.weight"], sd[e + "fc_layers.3.bias"])), 0.01)   # :58-60
    x = F.linear(x, sd[e + "fc_layers.6.weight"], sd[e + "fc_layers.6.bias"])                         # :61
    mean = F.linear(x, sd[e + "fc_mu.weight"], sd[e + "fc_mu.bias"])                                  # :80
    log_var = F.linear(x, sd[e + "fc_logvar.weight"], sd[e + "fc_logvar.bias"])                       # :81
    return mean, log_var


def window_latents(sd: SD, coeffs_seq: torch.Tensor, window_step_size: int, padding: int = 0) -> torch.Tensor:
    """test_evaluate.py:89-95: latent mean of every sliding window of one (T, 32) sequence -> (num_windows, 64)."""
    T = coeffs_seq.shape[0]
    n = (T - SEQ_LEN) // window_step_size + 1 - padding
    out = []
    for w in range(max(n, 0)):
        s = window_step_size * w
        out.append(encode(sd, coeffs_seq[None, s:s + SEQ_LEN])[0][0])
    return torch.stack(out) if out else torch.empty(0, 64)
