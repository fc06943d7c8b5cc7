"""Pin the CPU oracle to the golden vectors captured from the reference's own
modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import pipeline as op
from oracle import unet as ou
from oracle import wav2vec2 as ow
from said_amd.util import synth

torch.set_grad_enabled(False)
TOL = dict(rtol=2e-5, atol=2e-5)


def test_weight_fill_matches_fixture(golden, unet_sd, w2v_sd):
    assert synth.state_dict_checksum(unet_sd) == float(golden("weights_checksum")["unet"])
    assert synth.state_dict_checksum(w2v_sd) == float(golden("g5_wav2vec2")["checksum"])


def test_g1_timestep_embedding(golden):
    g = golden("g1_timestep_embedding")
    out = ou.timestep_embedding(torch.from_numpy(g["t"]), 192).numpy()
    np.testing.assert_array_equal(out, g["emb"])


@pytest.mark.parametrize("T,S", [(8, 8), (600, 600), (1800, 1800), (600, 499), (7, 10), (48, 48), (180, 180), (10, 7)])
def test_g2_alignment_band(golden, T, S):
    g = golden("g2_alignment_band")
    m = ~ou.alignment_mask(1, T, S)[0]
    lo = m.float().argmax(dim=1).numpy()
    hi = lo + m.sum(dim=1).numpy()
    np.testing.assert_array_equal(lo, g[f"lo_{T}_{S}"])
    np.testing.assert_array_equal(hi, g[f"hi_{T}_{S}"])
    if T == S:  # band is {i-1, i, i+1} whenever S == T
        i = np.arange(T)
        np.testing.assert_array_equal(lo, np.maximum(i - 1, 0))
        np.testing.assert_array_equal(hi, np.minimum(i + 2, S))


def test_g3_blocks(golden, unet_sd):
    g = golden("g3_blocks")
    x192 = synth.synth_latents(11, (2, 192, 48))
    x384 = synth.synth_latents(12, (2, 384, 48))
    emb = synth.synth_latents(13, (2, 768))
    ctx = synth.synth_latents(14, (2, 48, 768))
    np.testing.assert_allclose(ou.res_block(unet_sd, "model.input_blocks.1.0", x192, emb).numpy(), g["res192"], **TOL)
    np.testing.assert_allclose(ou.res_block(unet_sd, "model.output_blocks.0.0", x384, emb).numpy(), g["res384"], **TOL)
    np.testing.assert_allclose(ou.spatial_transformer(unet_sd, "model.input_blocks.1.1", x192, ctx).numpy(), g["st"], **TOL)
    np.testing.assert_allclose(ou.time_embed(unet_sd, torch.tensor([3, 977])).numpy(), g["time_embed"], **TOL)


@pytest.mark.parametrize("B,T,seed", [(1, 48, 21), (2, 48, 22), (1, 600, 23), (2, 600, 24), (2, 37, 25)])
def test_g4_unet(golden, unet_sd, B, T, seed):
    g = golden("g4_unet")
    x = synth.synth_latents(seed, (B, T, 32))
    c = synth.synth_latents(seed + 100, (B, T, 768))
    ts = torch.tensor([999, 17][:B])
    out = ou.unet1d_forward(unet_sd, x, ts, c).numpy()
    ref = g[f"out_B{B}_T{T}"]
    assert np.abs(ref).mean() > 0.05  # zero_module layers really were re-randomised
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4)


def test_g4_unet_context_len_differs(golden, unet_sd):
    g = golden("g4_unet")
    x = synth.synth_latents(26, (1, 40, 32))
    c = synth.synth_latents(126, (1, 25, 768))
    out = ou.unet1d_forward(unet_sd, x, torch.tensor([321]), c).numpy()
    np.testing.assert_allclose(out, g["out_B1_T40_S25"], rtol=1e-4, atol=1e-4)


def test_g5_wav2vec2(golden, w2v_sd):
    g = golden("g5_wav2vec2")
    proc = op.process_audio(synth.synth_waveform(0, 16000))
    lhs, feats = ow.wav2vec2_forward(w2v_sd, proc, 60)
    np.testing.assert_allclose(feats.numpy(), g["conv_feats"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lhs.numpy(), g["last_hidden_state"], rtol=2e-4, atol=2e-4)
    proc2 = op.process_audio([synth.synth_waveform(1, 8000).numpy(), synth.synth_waveform(2, 8000).numpy()])
    np.testing.assert_allclose(ow.wav2vec2_forward(w2v_sd, proc2, 30)[0].numpy(), g["lhs_b2_f30"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ow.wav2vec2_forward(w2v_sd, proc2[:1], None)[0].numpy(), g["lhs_noint"], rtol=2e-4, atol=2e-4)


def test_g6_process_audio(golden):
    g = golden("g6_process_audio")
    wav6 = synth.synth_waveform(5, 4000) * 3.0 + 0.25
    np.testing.assert_array_equal(op.process_audio(wav6).numpy(), g["out"])
    np.testing.assert_array_equal(op.process_audio(wav6.numpy()).numpy(), g["out"])
    both = op.process_audio([wav6.numpy(), synth.synth_waveform(6, 4000).numpy()])
    np.testing.assert_array_equal(both.numpy(), g["out_list"])


def test_g7_fit_audio_unet(golden):
    for n, fps, div, n_fit, win in golden("g7_fit_audio")["rows"]:
        w = torch.arange(int(n), dtype=torch.float32)
        wf, wl = op.fit_audio_unet(w, 16000, int(fps), int(div))
        assert wf.shape[0] == n_fit and wl == win
        assert torch.equal(wf[: int(n)], w) and float(wf[int(n):].abs().sum()) == 0.0


def test_g8_csv_reader(golden, tmp_path):
    g = golden("g8_csv")
    p = tmp_path / "o.csv"
    p.write_bytes(g["text"].tobytes())
    back = op.load_blendshape_coeffs(str(p)).numpy()
    np.testing.assert_array_equal(back, g["back"])
    header = g["text"].tobytes().decode().splitlines()[0].split(",")
    assert header == op.BLENDSHAPE_CLASSES and len(header) == 32


# ---------------------------------------------------------------- G9: loop control flow (scheduler leg unpinned)
from g9_cases import G9_CASES, g9_inputs  # noqa: E402


@pytest.mark.parametrize("name", list(G9_CASES))
def test_g9_loop_control_flow_vs_reference_inference(golden, name):
    """oracle.pipeline.inference against the reference's own SAID_UNet1D.inference (diffusion.py:308-472) captured on
    CPU with ``diffusers`` stubbed by oracle/scheduler.py: pins the loop control flow, NOT the scheduler arithmetic
    (same restatement on both sides).  Differences are fp32 summation order between the reference's UNet / encoder
    and the oracle's restatement of them, amplified over the chain."""
    c = G9_CASES[name]
    g = golden("g9_loop_control_flow_scheduler_leg_unpinned")
    proc, kw, noise, init_t = g9_inputs(c)
    sd = synth.said_state_dict(num_w2v_layers=2)
    out = op.inference(sd, proc, prediction_type=c.get("pred", "epsilon"), latent_scale=c.get("latent_scale", 1.0), **kw, **noise)
    ref = g[name + "_result"]
    err = float(np.abs(out.result.numpy() - ref).max())
    print(f"g9 {name}: max abs err vs reference loop {err:.2e}")
    assert out.result.shape == ref.shape and err <= 5e-4   # measured 1e-6 .. 1e-4 (fp32 summation-order noise amplified over the chain)
    if c.get("save_intermediate", False):
        ri = g[name + "_inter"]
        assert len(out.intermediates) == ri.shape[0] == init_t
        assert float(np.abs(torch.stack(out.intermediates).numpy() - ri).max()) <= 2e-3


# ---------------------------------------------------------------- G10: VAE encoder
VAE_PTH = "/root/reference/model/vae.pth"   # the reference's trained weights: build container only


def test_g10_vae_encoder_synth_weights(golden):
    """oracle/vae.py against the reference's own BCVAE.encode (eval mode) with the deterministic weight fill."""
    import os
    from oracle import vae as ov
    g = golden("g10_vae_encoder")
    sd = synth.vae_encoder_state_dict()
    coeffs = torch.sigmoid(synth.synth_latents(41, (5, 120, 32)))
    seq = torch.sigmoid(synth.synth_latents(42, (300, 32)))
    mean, logvar = ov.encode(sd, coeffs)
    np.testing.assert_allclose(mean.numpy(), g["synth_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logvar.numpy(), g["synth_log_var"], rtol=1e-5, atol=1e-6)
    for step, pad in ((10, 0), (1, 3)):
        got = ov.window_latents(sd, seq, step, pad).numpy()
        ref = g[f"synth_win_s{step}_p{pad}"]
        assert got.shape == ref.shape == ((300 - 120) // step + 1 - pad, 64)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)


@pytest.mark.skipif(not __import__("os").path.exists(VAE_PTH), reason="model/vae.pth lives in /root/reference (build container only)")
def test_g10_vae_encoder_reference_weights(golden):
    from oracle import vae as ov
    g = golden("g10_vae_encoder")
    sd = torch.load(VAE_PTH, map_location="cpu")
    coeffs = torch.sigmoid(synth.synth_latents(41, (5, 120, 32)))
    mean, logvar = ov.encode(sd, coeffs)
    np.testing.assert_allclose(mean.numpy(), g["real_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logvar.numpy(), g["real_log_var"], rtol=1e-5, atol=1e-6)


def test_bf16_emulating_mode_of_the_oracle(unet_sd):
    """oracle/unet.py ROUND_OPERANDS (test infrastructure for bf16 mode, tests/test_gpu_round5.py): off by default and reversible (the golden-pinned path is untouched);
    on operands that ARE bf16 numbers the rounded product is the exact one; the restated online softmax equals the plain one up to the rounding of the probabilities
    and does not depend on how the keys are sliced beyond that."""
    import torch.nn.functional as F
    x = synth.synth_latents(5, (1, 40, 32)); c = synth.synth_latents(6, (1, 40, 768)); ts = torch.tensor([321])
    ref = ou.unet1d_forward(unet_sd, x, ts, c)
    try:
        ou.ROUND_OPERANDS = "bf16"
        emu = ou.unet1d_forward(unet_sd, x, ts, c)
        a = synth.synth_latents(7, (3, 17, 192)).to(torch.bfloat16).float()
        w = synth.synth_latents(8, (1, 48, 192))[0].to(torch.bfloat16).float()
        assert torch.equal(ou._linear(a, w), F.linear(a.double(), w.double()).float())
    finally:
        ou.ROUND_OPERANDS = None
    assert torch.equal(ou.unet1d_forward(unet_sd, x, ts, c), ref)
    d = float((emu - ref).abs().max()) / float(ref.abs().max())
    assert 1e-4 < d < 2e-2, d     # bf16 roundings: percent-level on this random-weight network, never zero
    q, k, v = (synth.synth_latents(10 + i, (4, 100, 32)) for i in range(3))
    r = lambda t: t.to(torch.bfloat16).float()
    plain = torch.einsum("bij,bjd->bid", (torch.einsum("bid,bjd->bij", r(q), r(k)) * 32 ** -0.5).softmax(dim=-1), r(v))
    o1, o4 = ou.attn_bf16_online(q, k, v, 1), ou.attn_bf16_online(q, k, v, 4)
    assert float((o1 - plain).abs().max()) < 2e-2 and float((o4 - o1).abs().max()) < 2e-2 and float((o4 - plain).abs().max()) > 0
