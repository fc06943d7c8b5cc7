"""VAE encoder on the MI355X through the C ABI (said_vae_*) against golden G10 — the reference's
own BCVAE.encode in eval mode — and the CPU oracle.  Tolerance: 1e-4 relative to the output range."""
import os

import numpy as np
import pytest
import torch

from oracle import vae as ov
from said_amd.util import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
VAE_PTH = os.environ.get("SAID_VAE_PTH", "/root/reference/model/vae.pth")


def _model(dev, sd_enc):
    from said_amd.model.vae import BCVAE
    m = BCVAE()
    full = m.state_dict()
    full.update(sd_enc)
    m.load_state_dict(full, strict=True)
    return m.to(dev).eval()


def test_vae_encode_vs_golden_and_oracle(golden):
    dev = torch.device("cuda:0")
    g = golden("g10_vae_encoder")
    sd = synth.vae_encoder_state_dict()
    m = _model(dev, sd)
    coeffs = torch.sigmoid(synth.synth_latents(41, (5, 120, 32)))
    lat = m.encode(coeffs.to(dev))
    for got, ref in ((lat.mean, g["synth_mean"]), (lat.log_var, g["synth_log_var"])):
        err = np.abs(got.cpu().numpy() - ref).max() / np.abs(ref).max()
        print(f"vae encode: max err {err:.2e} of range")
        assert err <= 1e-4
    assert any("libsaid_hip.so" in ln for ln in open("/proc/self/maps"))


@pytest.mark.parametrize("step,pad", [(10, 0), (1, 3)])
def test_vae_sliding_windows_vs_golden(golden, step, pad):
    """generate_latents_info's window loop (test_evaluate.py:89-95) as one engine call over the (T, 32) sequence."""
    dev = torch.device("cuda:0")
    g = golden("g10_vae_encoder")
    m = _model(dev, synth.vae_encoder_state_dict())
    seq = torch.sigmoid(synth.synth_latents(42, (300, 32)))
    got = m.encode_windows(seq.to(dev), step, pad).cpu().numpy()
    ref = g[f"synth_win_s{step}_p{pad}"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


def test_vae_large_window_count_and_edge_cases():
    """More windows than one 4096-window chunk; a ragged count; zero windows; wrong shape -> error, not garbage."""
    dev = torch.device("cuda:0")
    sd = synth.vae_encoder_state_dict()
    m = _model(dev, sd)
    seq = torch.sigmoid(synth.synth_latents(43, (4500 + 119, 32)))
    got = m.encode_windows(seq.to(dev), 1).cpu()
    assert got.shape == (4500, 64)
    for w in (0, 1, 4095, 4096, 4499):
        ref = ov.encode(sd, seq[None, w:w + 120])[0][0]
        assert float((got[w] - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert m.encode_windows(seq[:100].to(dev), 10).shape == (0, 64)
    with pytest.raises(ValueError):
        m.encode(torch.zeros(2, 100, 32, device=dev))
    m.train()
    from said_amd import _engine
    with pytest.raises(_engine.EngineError):
        m.encode(torch.zeros(1, 120, 32, device=dev))


@pytest.mark.skipif(not os.path.exists(VAE_PTH), reason="the reference's trained vae.pth is not on the GPU box (set SAID_VAE_PTH)")
def test_vae_reference_weights(golden):
    dev = torch.device("cuda:0")
    from said_amd.model.vae import BCVAE
    m = BCVAE()
    m.load_state_dict(torch.load(VAE_PTH, map_location="cpu"), strict=True)
    m.to(dev).eval()
    g = golden("g10_vae_encoder")
    coeffs = torch.sigmoid(synth.synth_latents(41, (5, 120, 32)))
    lat = m.encode(coeffs.to(dev))
    assert np.abs(lat.mean.cpu().numpy() - g["real_mean"]).max() <= 1e-4 * np.abs(g["real_mean"]).max()
