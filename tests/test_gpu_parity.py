"""Parity tests proper (run on the MI355X with `-m gpu`): the HIP path through the C ABI
versus the CPU oracle and the committed golden vectors.

Stated tolerances: single UNet evaluation <= 1e-4 of the output range;
scheduler arithmetic bit-exact given identical eps; end-to-end result <= 1e-3 abs on the
[0,1]-clamped coefficients (5e-3 after long stochastic chains, stated per test)."""
import numpy as np
import pytest
import torch

from oracle import pipeline as op
from oracle import scheduler as osch
from oracle import unet as ou
from oracle import wav2vec2 as ow
from said_amd.util import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    from said_amd.model.diffusion import SAID_UNet1D
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    return m


@pytest.fixture(scope="module")
def sd_full():
    return synth.said_state_dict()


def test_engine_library_is_loaded(model, dev):
    import ctypes
    from said_amd import _engine
    assert isinstance(_engine.load_library(), ctypes.CDLL)
    model._get_engine(2, 64)
    assert any("libsaid_hip.so" in ln for ln in open("/proc/self/maps"))


def test_cpu_device_is_refused():
    from said_amd import _engine
    from said_amd.model.diffusion import SAID_UNet1D
    m = SAID_UNet1D()
    with pytest.raises(_engine.EngineError):
        m.inference(torch.zeros(1, 1600))


# ---------------------------------------------------------------- UNet (SAID.forward)
@pytest.mark.parametrize("B,T,seed", [(1, 48, 21), (2, 48, 22), (1, 600, 23), (2, 600, 24), (2, 37, 25)])
def test_unet_forward_vs_golden_and_oracle(golden, unet_sd, model, dev, B, T, seed):
    x = synth.synth_latents(seed, (B, T, 32))
    c = synth.synth_latents(seed + 100, (B, T, 768))
    ts = torch.tensor([999, 17][:B])
    out = model(x.to(dev), ts.to(dev), c.to(dev)).cpu().numpy()
    ref = golden("g4_unet")[f"out_B{B}_T{T}"]  # captured from the reference's own UNet
    scale = np.abs(ref).max()
    assert np.abs(out - ref).max() <= 1e-4 * scale
    orc = ou.unet1d_forward(unet_sd, x, ts, c).numpy()
    assert np.abs(out - orc).max() <= 1e-4 * scale


def test_unet_forward_context_length_differs(golden, model, dev):
    x = synth.synth_latents(26, (1, 40, 32))
    c = synth.synth_latents(126, (1, 25, 768))
    out = model(x.to(dev), torch.tensor([321]).to(dev), c.to(dev)).cpu().numpy()
    ref = golden("g4_unet")["out_B1_T40_S25"]
    assert np.abs(out - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("B,T,S", [(1, 7, 10), (1, 31, 31), (3, 33, 33), (2, 65, 50), (1, 100, 77), (5, 96, 96), (4, 160, 201),
                                   (9, 64, 64), (2, 10, 100), (3, 37, 400), (1, 5, 333)])
def test_unet_forward_ragged_shapes_vs_oracle(model, unet_sd, dev, B, T, S):
    """Frame counts that are not multiples of the 32-token tile, fewer frames than one tile, context lengths shorter and
    longer than the frame count (general alignment windows — the last three cases have windows of 12, 13 and 69 keys: wider than the
    eight the fused band epilogue holds, served by the generic band kernel since round 4), odd batch sizes, and the batch sizes at
    which the tile shapes change (NB=1 -> 2 at 64 token tiles)."""
    x = synth.synth_latents(300 + T, (B, T, 32))
    c = synth.synth_latents(400 + S, (B, S, 768))
    ts = (torch.arange(B) * 137 + 11) % 1000
    out = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    ref = ou.unet1d_forward(unet_sd, x, ts, c)
    assert float((out - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_unet_scalar_and_single_timestep_broadcast(model, unet_sd, dev):
    x = synth.synth_latents(5, (3, 33, 32))
    c = synth.synth_latents(6, (3, 33, 768))
    ref = ou.unet1d_forward(unet_sd, x, torch.tensor([77, 77, 77]), c).numpy()
    for t in (torch.tensor(77), torch.tensor([77])):
        out = model(x.to(dev), t.to(dev), c.to(dev)).cpu().numpy()
        assert np.abs(out - ref).max() <= 1e-4 * np.abs(ref).max()


def test_unet_large_batch_tiling_path(model, unet_sd, dev):
    """Be*T large enough to take the NB=6/KS=4 workgroup shape; same numbers expected."""
    B, T = 24, 352
    x = synth.synth_latents(41, (B, T, 32))
    c = synth.synth_latents(42, (B, T, 768))
    ts = torch.arange(B) * 41 % 1000
    out = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    ref = ou.unet1d_forward(unet_sd, x[:3], ts[:3], c[:3])
    assert float((out[:3] - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    ref_last = ou.unet1d_forward(unet_sd, x[-1:], ts[-1:], c[-1:])
    assert float((out[-1:] - ref_last).abs().max()) <= 1e-4 * float(ref_last.abs().max())


# ---------------------------------------------------------------- scheduler arithmetic
@pytest.mark.parametrize("pred", ["epsilon", "sample", "v_prediction"])
@pytest.mark.parametrize("N,eta", [(1000, 0.0), (50, 0.0), (100, 1.0), (7, 0.5)])
def test_scheduler_step_bit_exact(model, dev, pred, N, eta):
    from said_amd.scheduler import DDIMScheduler
    eng = model._get_engine(2, 64)
    sch = DDIMScheduler(prediction_type=pred)
    sch._engine = eng
    sch.set_timesteps(N)
    o = osch.OracleDDIM(1000, pred)
    o.set_timesteps(N)
    assert torch.equal(sch.timesteps, o.timesteps)
    assert torch.equal(sch.alphas_cumprod, o.alphas_cumprod)
    g = torch.Generator().manual_seed(N)
    for t in [int(sch.timesteps[0]), int(sch.timesteps[len(sch.timesteps) // 2]), int(sch.timesteps[-1])]:
        eps = torch.randn(2, 19, 32, generator=g)
        x = torch.randn(2, 19, 32, generator=g) * 1.5
        nz = torch.randn(2, 19, 32, generator=g)
        ref = o.step(eps, t, x, eta=eta, variance_noise=nz if eta > 0 else None)
        got = sch.step(eps.to(dev), t, x.to(dev), eta=eta, variance_noise=nz.to(dev) if eta > 0 else None).prev_sample.cpu()
        assert torch.equal(got, ref), f"t={t}: max diff {(got - ref).abs().max()}"


def test_add_noise_velocity_cfg_and_blend_bit_exact(model, dev):
    eng = model._get_engine(2, 64)
    sch = model.noise_scheduler
    sch.set_timesteps(100)
    o = osch.OracleDDIM(1000, "epsilon")
    o.set_timesteps(100)
    g = torch.Generator().manual_seed(3)
    x, n = torch.randn(3, 11, 32, generator=g), torch.randn(3, 11, 32, generator=g)
    ts = torch.tensor([990, 500, 0])
    assert torch.equal(sch.add_noise(x.to(dev), n.to(dev), ts).cpu(), o.add_noise(x, n, ts))
    assert torch.equal(sch.get_velocity(x.to(dev), n.to(dev), ts).cpu(), o.get_velocity(x, n, ts))
    # CFG combine + step + mask blend through said_ddim_step
    e_c, e_u = torch.randn(3, 11, 32, generator=g), torch.randn(3, 11, 32, generator=g)
    init, mask = torch.rand(3, 11, 32, generator=g), (torch.rand(3, 11, 32, generator=g) > 0.5).float()
    t, t_next = 500, 490
    row = sch._coef_row(t, 0.0, t_next)
    got = eng.ddim_step(e_c.to(dev), x.to(dev), row, "epsilon", eps_uncond=e_u.to(dev), guidance_scale=2.0,
                        init_latents=init.to(dev), edit_noise=n.to(dev), mask=mask.to(dev)).cpu()
    eps = e_c + 2.0 * (e_c - e_u)
    prev = o.step(eps, t, x)
    ref = o.add_noise(init, n, torch.tensor(t_next)) * mask + prev * (1 - mask)
    assert torch.equal(got, ref)


def test_scheduler_self_consistency():
    """Unpinned scheduler: properties that must hold for any correct DDIM restatement."""
    o = osch.OracleDDIM(1000, "epsilon")
    ac = o.alphas_cumprod
    assert bool((ac[1:] < ac[:-1]).all()) and 0 < float(ac[-1]) < float(ac[0]) < 1
    o.set_timesteps(1000)
    assert o.timesteps.tolist() == list(range(999, -1, -1))
    o.set_timesteps(50)
    assert o.timesteps.tolist() == list(range(980, -1, -20))
    # feeding the true noise returns sqrt(a_prev) x0 + sqrt(1 - a_prev) eps (x0 inside the clip range)
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand(2, 5, 32, generator=g) * 1.6 - 0.8
    eps = torch.randn(2, 5, 32, generator=g)
    t = 500
    xt = o.add_noise(x0, eps, torch.tensor([t, t]))
    prev = o.step(eps, t, xt)
    want = o.add_noise(x0, eps, torch.tensor([t - 20, t - 20]))
    assert float((prev - want).abs().max()) < 1e-5
    # eta = 1, N = 1000: variance equals the DDPM posterior variance
    o.set_timesteps(1000)
    v = o._get_variance(500, 499)
    beta_t = 1 - ac[500] / ac[499]
    assert abs(float(v) - float((1 - ac[499]) / (1 - ac[500]) * beta_t)) < 1e-7


# ---------------------------------------------------------------- audio encoder
# measured on MI355X (round 2): 2.9e-6 .. 4.1e-6 abs over the five encodes (|ref| max 3.8): bound = 4e-6 x max(1, |ref|max), i.e. 3-4x the measured error
AUDIO_TOL = 4e-6

def test_audio_encoder_vs_golden(golden, model, w2v_sd, dev):
    g = golden("g5_wav2vec2")
    proc = model.process_audio(synth.synth_waveform(0, 16000))
    assert torch.equal(proc, op.process_audio(synth.synth_waveform(0, 16000)))
    lhs = model.get_audio_embedding(proc.to(dev), 60).cpu().numpy()
    ref = g["last_hidden_state"]  # captured from the reference's ModifiedWav2Vec2Model
    print(f"audio 1 s vs golden: max abs err {np.abs(lhs - ref).max():.3e}")
    assert np.abs(lhs - ref).max() <= AUDIO_TOL * max(1.0, np.abs(ref).max())
    proc2 = model.process_audio([synth.synth_waveform(1, 8000).numpy(), synth.synth_waveform(2, 8000).numpy()])
    lhs2 = model.get_audio_embedding(proc2.to(dev), 30).cpu().numpy()
    print(f"audio 2 x 0.5 s vs golden: max abs err {np.abs(lhs2 - g['lhs_b2_f30']).max():.3e}")
    assert np.abs(lhs2 - g["lhs_b2_f30"]).max() <= AUDIO_TOL * max(1.0, np.abs(g["lhs_b2_f30"]).max())
    lhs3 = model.audio_encoder(proc2[:1].to(dev), num_frames=None).last_hidden_state.cpu().numpy()
    assert lhs3.shape == g["lhs_noint"].shape
    print(f"audio no-interp vs golden: max abs err {np.abs(lhs3 - g['lhs_noint']).max():.3e}")
    assert np.abs(lhs3 - g["lhs_noint"]).max() <= AUDIO_TOL * max(1.0, np.abs(g["lhs_noint"]).max())


def test_audio_encoder_3s_vs_oracle(model, w2v_sd, dev):
    proc = op.process_audio(synth.synth_waveform(7, 48000))
    ref = ow.wav2vec2_forward(w2v_sd, proc, 180)[0].numpy()
    got = model.get_audio_embedding(proc.to(dev), 180).cpu().numpy()
    print(f"audio 3 s vs oracle: max abs err {np.abs(got - ref).max():.3e}")
    assert np.abs(got - ref).max() <= AUDIO_TOL * max(1.0, np.abs(ref).max())


# ---------------------------------------------------------------- full loop
def _loop_case(model, sd_full, dev, *, B, Ta, N, gs, eta=0.0, rescale=0.0, pred="epsilon", edit=False, strength=1.0,
               save_intermediate=False, tol=1e-3):
    T = int(Ta / 16000 * 60)
    wav = torch.stack([synth.synth_waveform(10 + i, Ta) for i in range(B)])
    proc = op.process_audio([w.numpy() for w in wav])
    init_lat = synth.synth_latents(100, (B, T, 32))
    kw = {}
    okw = {}
    if edit:
        init_samples = torch.sigmoid(synth.synth_latents(101, (B, T, 32))) * 0.5
        mask = torch.zeros(B, T, 32)
        mask[:, : T // 3] = 1.0
        mask[:, :, :4] = 1.0
        en = synth.synth_latents(102, (B, T, 32))
        kw = dict(init_samples=init_samples.to(dev), mask=mask.to(dev), edit_noise=en.to(dev))
        okw = dict(init_samples=init_samples, mask=mask, edit_noise=en)
    init_t = min(int(N * strength), N)
    sn = synth.synth_latents(103, (init_t, B, T, 32)) if eta > 0 else None
    model.noise_scheduler.config.prediction_type = pred
    try:
        out = model.inference(proc.to(dev), num_inference_steps=N, strength=strength, guidance_scale=gs, guidance_rescale=rescale,
                              eta=eta, init_latents=init_lat.to(dev), step_noise=None if sn is None else sn.to(dev),
                              save_intermediate=save_intermediate, **kw)
    finally:
        model.noise_scheduler.config.prediction_type = "epsilon"
    ref = op.inference(sd_full, proc, init_latents=init_lat, num_inference_steps=N, strength=strength, guidance_scale=gs,
                       guidance_rescale=rescale, eta=eta, prediction_type=pred, step_noise=sn, save_intermediate=save_intermediate, **okw)
    got = out.result.cpu()
    assert got.shape == (B, T, 32) and float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    err = float((got - ref.result).abs().max())
    assert err <= tol, f"end-to-end max abs err {err}"
    if save_intermediate:
        assert len(out.intermediates) == len(ref.intermediates) == init_t
        for a, b in zip(out.intermediates, ref.intermediates):
            assert float((a.cpu() - b).abs().max()) <= 10 * tol
    return got, ref.result


def test_loop_cfg_1s_50steps(model, sd_full, dev):
    _loop_case(model, sd_full, dev, B=2, Ta=16000, N=50, gs=2.0)


def test_loop_no_guidance_intermediates(model, sd_full, dev):
    _loop_case(model, sd_full, dev, B=1, Ta=16000, N=20, gs=1.0, save_intermediate=True)


def test_loop_eta_and_rescale(model, sd_full, dev):
    _loop_case(model, sd_full, dev, B=2, Ta=8000, N=25, gs=2.5, eta=1.0, rescale=0.7, tol=2e-3)


@pytest.mark.parametrize("pred", ["sample", "v_prediction"])
def test_loop_prediction_types(model, sd_full, dev, pred):
    _loop_case(model, sd_full, dev, B=1, Ta=8000, N=10, gs=2.0, pred=pred)


def test_loop_editing_mask_strength(model, sd_full, dev):
    got, ref = _loop_case(model, sd_full, dev, B=2, Ta=16000, N=30, gs=2.0, edit=True, strength=0.6, save_intermediate=True)


def test_loop_zero_steps_returns_clamped_start(model, dev):
    proc = op.process_audio(synth.synth_waveform(0, 8000))
    lat = synth.synth_latents(1, (1, 30, 32))
    init = torch.rand(1, 30, 32)
    out = model.inference(proc.to(dev), init_samples=init.to(dev), num_inference_steps=10, strength=0.0, guidance_scale=2.0,
                          edit_noise=lat.to(dev))
    # strength 0 -> init_timestep 0 -> timesteps[-0] is timesteps[0]; no loop steps run (diffusion.py:373-409)
    sch = osch.OracleDDIM()
    sch.set_timesteps(10)
    want = sch.add_noise(init, lat, torch.tensor([int(sch.timesteps[0])])).clamp(0, 1)
    assert torch.equal(out.result.cpu(), want)


def test_loop_is_deterministic_and_graph_replayed(model, dev):
    proc = op.process_audio(synth.synth_waveform(3, 16000)).to(dev)
    lat = synth.synth_latents(9, (1, 60, 32)).to(dev)
    a = model.inference(proc, num_inference_steps=12, guidance_scale=2.0, init_latents=lat).result
    b = model.inference(proc, num_inference_steps=12, guidance_scale=2.0, init_latents=lat).result
    assert torch.equal(a, b)
    assert model._eng.graph_num_nodes() >= 20  # one captured graph covers the whole step (24 launches at this batch since round 5's fused transformer tail; 40 before)


def test_idempotent_clamp_and_linearity_properties_full_size(model, dev):
    """BASELINE cfg2 size (T=600): size-independent properties instead of an oracle run."""
    T = 600
    x = synth.synth_latents(51, (1, T, 32)).to(dev)
    c = synth.synth_latents(52, (1, T, 768)).to(dev)
    t = torch.tensor([400]).to(dev)
    e1 = model(x, t, c)
    e2 = model(x, t, c)
    assert torch.equal(e1, e2)  # fixed reduction order => bit-reproducible
    # batch independence: evaluating a sample alone or inside a batch gives the same numbers up to fp32 summation
    # order (the tile shapes — and with them the order in which K is accumulated — depend on the batch size)
    xb = torch.cat([x, x.flip(1)])
    cb = torch.cat([c, c.flip(1)])
    eb = model(xb, torch.tensor([400, 400]).to(dev), cb)
    assert float((eb[:1] - e1).abs().max()) <= 2e-5 * float(e1.abs().max())
    # time-reversal equivariance of the whole network (convs are not symmetric, so only check finiteness + scale)
    assert torch.isfinite(eb).all() and float(eb.abs().max()) < 1e3


# ---------------------------------------------------------------- BASELINE.json configs 3-5 shapes
def test_unet_forward_long_sequence_vs_oracle(model, unet_sd, dev):
    """cfg5 shape: T = S = 1800 (30 s): 57 key tiles per attention workgroup, 57 GroupNorm partials per channel,
    alignment band over 1800 audio tokens."""
    T = 1800
    x = synth.synth_latents(61, (1, T, 32))
    c = synth.synth_latents(62, (1, T, 768))
    ts = torch.tensor([731])
    out = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    ref = ou.unet1d_forward(unet_sd, x, ts, c)
    assert float((out - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_loop_editing_30s_in_betweening(model, sd_full, dev):
    """cfg5: editing mode on 30 s of audio (T=1800), in-betweening mask (middle third regenerated) + 4 pinned channels,
    a few DDIM steps against the oracle (the audio encoder runs on 480k samples: 1499 conv frames -> 1800)."""
    B, Ta, N = 1, 480000, 3
    T = 1800
    wav = synth.synth_waveform(77, Ta)
    proc = op.process_audio([wav.numpy()])
    init_lat = synth.synth_latents(110, (B, T, 32))
    init_samples = torch.sigmoid(synth.synth_latents(111, (B, T, 32))) * 0.5
    mask = torch.zeros(B, T, 32)
    mask[:, :600] = 1.0
    mask[:, 1200:] = 1.0
    mask[:, :, :4] = 1.0
    en = synth.synth_latents(112, (B, T, 32))
    out = model.inference(proc.to(dev), num_inference_steps=N, guidance_scale=2.0, init_latents=init_lat.to(dev),
                          init_samples=init_samples.to(dev), mask=mask.to(dev), edit_noise=en.to(dev))
    ref = op.inference(sd_full, proc, init_latents=init_lat, num_inference_steps=N, guidance_scale=2.0,
                       init_samples=init_samples, mask=mask, edit_noise=en)
    got = out.result.cpu()
    assert got.shape == (B, T, 32)
    assert float((got - ref.result).abs().max()) <= 1e-3
    # masked-in frames keep the (re-noised then denoised) init trajectory: at the last step the blend uses the clean init
    assert float((got[:, :600] - init_samples[:, :600].clamp(0, 1)).abs().max()) <= 1e-6


def test_batch32_matches_single_clip_runs(model, dev):
    """cfg3/cfg4 shape (32 clips per GPU, UNet batch 64 under guidance, T=600): every clip of the batch must come out
    as if it had been run alone (no cross-sample op exists on the path).  Large batches take different
    tile shapes, so equality is up to fp32 summation order."""
    B, T, N = 32, 600, 2
    ctx = synth.synth_latents(120, (B, T, 768)).to(dev)
    lat = synth.synth_latents(121, (B, T, 32)).to(dev)
    wav = torch.zeros(B, T * 16000 // 60, device=dev)   # only its shape is used when the embedding is injected
    big = model.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result
    for i in (0, 17, 31):
        one = model.inference(wav[i:i + 1], audio_embedding=ctx[i:i + 1], num_inference_steps=N, guidance_scale=2.0, init_latents=lat[i:i + 1]).result
        assert float((big[i:i + 1] - one).abs().max()) <= 2e-5


# ---------------------------------------------------------------- bf16 multiplies (BASELINE.json configs[2])
def test_bf16_mfma_mode_error_vs_fp32_oracle(model, unet_sd, dev):
    """configs[2] precision: operands of the UNet GEMMs rounded to bf16, fp32 accumulation / statistics / storage.
    No bit-exactness claim: the error against the fp32 oracle is bounded and reported; the switch must
    actually change the arithmetic and must be reversible."""
    B, T = 2, 600
    x = synth.synth_latents(21, (B, T, 32))
    c = synth.synth_latents(121, (B, T, 768))
    ts = torch.tensor([999, 17])
    ref = ou.unet1d_forward(unet_sd, x, ts, c)
    scale = float(ref.abs().max())
    o32 = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    try:
        model.set_mfma_dtype("bf16")
        o16 = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
        assert model._eng.get_precision() == "bf16"
    finally:
        model.set_mfma_dtype("fp32")
    e16 = float((o16 - ref).abs().max()) / scale
    rms = float((o16 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"bf16 UNet: max err {e16:.2e} of range, rms rel {rms:.2e}")
    assert 1e-5 < e16 <= 2e-2 and rms <= 2e-2          # measured 5.8e-3 / 5.9e-3
    again = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    assert torch.equal(again, o32) and float((o32 - ref).abs().max()) <= 1e-4 * scale


def test_bf16_loop_cfg_batch32_shape(model, dev):
    """configs[2] shape (32 clips, guidance, T=600) in bf16 for a few steps against the same loop in fp32: finite,
    clamped, close (random-weight networks amplify rounding over long chains, so the chain is kept short)."""
    B, T, N = 32, 600, 3
    ctx = synth.synth_latents(130, (B, T, 768)).to(dev)
    lat = synth.synth_latents(131, (B, T, 32)).to(dev)
    wav = torch.zeros(B, T * 16000 // 60, device=dev)
    r32 = model.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result
    try:
        model.set_mfma_dtype("bf16")
        r16 = model.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result
    finally:
        model.set_mfma_dtype("fp32")
    assert torch.isfinite(r16).all() and float(r16.min()) >= 0 and float(r16.max()) <= 1
    d = (r16 - r32).abs()
    print(f"bf16 loop: max abs diff {float(d.max()):.3e}, mean {float(d.mean()):.3e}")
    assert float(d.mean()) <= 1.2e-2   # 3x the measured 4.0e-3 (3 free-running steps at B=32; chains amplify: see the teacher-forced test)


# ---------------------------------------------------------------- round 2: chains at the headline's real length
# A free-running 1000-step chain of this RANDOM-WEIGHT network is chaotic: the CPU oracle run twice, the second time with
# its start latents scaled by (1 + 1e-6), differs from itself by 3e-2 after 100 steps and by O(1) after 200 (of 1000; 1 s
# clip, guidance 2 — measured in the build container).  No implementation pair can meet an end-to-end bound
# there, so the headline chain length is checked TEACHER-FORCED: every one of the 1000 steps (its own timestep embedding
# row, coefficient row, noise slice) starts from the oracle's latents of that step and must land on the oracle's next
# latents; free-running comparisons stay at chain lengths where the oracle's own sensitivity is below the bound.
def _oracle_steps(sd_full, lat, emb, ts_seg, sch, gs, sn_seg):
    """`len(ts_seg)` guided steps of diffusion.py:411-443 on the CPU oracle from the given latents."""
    _, sd_u, null = op.split_state_dict(sd_full)
    T = lat.shape[1]
    ctx = torch.cat([null.repeat(1, T, 1), emb])
    for i, t in enumerate(ts_seg):
        t = int(t)
        pred = ou.unet1d_forward(sd_u, torch.cat([lat] * 2), torch.tensor([t, t]), ctx)
        e_u, e_c = pred.chunk(2)
        lat = sch.step(e_c + gs * (e_c - e_u), t, lat, eta=0.0 if sn_seg is None else 1.0, variance_noise=None if sn_seg is None else sn_seg[i])
    return lat


def _teacher_forced(model, sd_full, dev, *, N, eta, seg_lens, Ta=16000, gs=2.0, tol_single=2e-4, tol_seg=1e-3, starts=None, chain=True):
    """chain=True: the oracle runs the whole N-step chain once and every HIP segment starts from the oracle's latents of its
    first step.  chain=False: no full oracle chain (it is most of the test's time) — segment k starts from latents at step k's
    noise level, add_noise(x0, n, t_k) of a synthetic clean sample, and the oracle runs just that segment from the same latents."""
    B, T = 1, int(Ta / 16000 * 60)
    proc = op.process_audio([synth.synth_waveform(10, Ta).numpy()])
    sd_a, _, _ = op.split_state_dict(sd_full)
    emb = op.get_audio_embedding(sd_a, proc, T)
    lat0 = synth.synth_latents(100, (B, T, 32))
    sn = synth.synth_latents(103, (N, B, T, 32)) if eta > 0 else None
    o = osch.OracleDDIM()
    o.set_timesteps(N)
    if chain:
        ref = op.inference(sd_full, proc, init_latents=lat0, num_inference_steps=N, guidance_scale=gs, eta=eta, step_noise=sn,
                           audio_embedding=emb, save_intermediate=True)
        xs = ref.intermediates                         # xs[k] = latents entering step k
    else:
        x0 = torch.sigmoid(synth.synth_latents(104, (B, T, 32))) * 0.5
    eng = model._get_engine(2 * B, T)
    sch = model.noise_scheduler
    sch.set_timesteps(N)
    ts = sch.timesteps.numpy()
    coef = sch.coef_table(ts, float(eta))
    emb_d = emb.to(dev)
    worst = {}
    for m in seg_lens:
        ks = starts[m] if starts and m in starts else (range(0, N - m + 1) if m == 1 else range(0, N - m + 1, max(m, N // 20)))
        w = 0.0
        for k in ks:
            x_in = xs[k] if chain else o.add_noise(x0, lat0, torch.tensor([int(ts[k])]))
            res, latf, _ = eng.denoise_loop(latents=x_in.to(dev), context=emb_d, timesteps=ts[k:k + m], coef=coef[k:k + m],
                                            prediction_type="epsilon", guidance_scale=gs, guidance_rescale=0.0, latent_scale=1.0,
                                            step_noise=None if sn is None else sn[k:k + m].contiguous().to(dev))
            # note: coefficient rows 5/6 (mask blend with the NEXT timestep) are unused without a mask
            if chain:
                want = xs[k + m] if k + m < N else None
                if want is not None:
                    w = max(w, float((latf.cpu() - want).abs().max()))
                else:
                    w = max(w, float((res.cpu() - ref.result).abs().max()))
            else:
                want = _oracle_steps(sd_full, x_in, emb, ts[k:k + m], o, gs, None if sn is None else sn[k:k + m])
                w = max(w, float((latf.cpu() - want).abs().max()))
        worst[m] = w
    return worst


def test_loop_1000_steps_teacher_forced_vs_oracle(model, sd_full, dev):
    """BASELINE configs[1]'s chain: all 1000 DDIM steps (guidance 2, 1 s clip), each started from the oracle's latents;
    plus 10-step segments (the 10-steps-per-graph replay the headline runs) at 20 places along the chain."""
    worst = _teacher_forced(model, sd_full, dev, N=1000, eta=0.0, seg_lens=[1, 10])
    print(f"N=1000 eta=0 teacher-forced: worst single-step err {worst[1]:.3e}, worst 10-step-segment err {worst[10]:.3e}")
    assert worst[1] <= 2e-4 and worst[10] <= 1e-3


def test_loop_1000_steps_eta1_teacher_forced_vs_oracle(model, sd_full, dev):
    """Same with eta = 1 (ancestral sampling, one injected noise draw per step): ALL 1000 single steps along the oracle's own free-running
    chain and 10-step segments at 20 places (round 3 checked every fourth step from synthetic latents to stay inside the suite's time limit;
    with the oracle's thread count pinned in conftest.py the full chain costs seconds — ADVICE r3)."""
    worst = _teacher_forced(model, sd_full, dev, N=1000, eta=1.0, seg_lens=[1, 10])
    print(f"N=1000 eta=1 teacher-forced: worst single-step err {worst[1]:.3e}, worst 10-step-segment err {worst[10]:.3e}")
    assert worst[1] <= 2e-4 and worst[10] <= 1e-3


def test_loop_headline_length_10_step_segments_vs_oracle(model, sd_full, dev):
    """The headline's own shape: T = 600 (10 s), guidance 2, the 1000-step schedule — 10-step segments (one replay of
    the ten-step graph) at an early, a middle and the last position of the chain, each from latents at that step's noise level,
    against the oracle's ten steps from the same latents.  (The 1 s chain above walks all 1000 steps; this pins the multi-step
    replay, step counter and coefficient rows at the benchmarked length.)"""
    worst = _teacher_forced(model, sd_full, dev, N=1000, eta=0.0, seg_lens=[10], Ta=160000, starts={10: [0, 500, 990]}, chain=False)
    print(f"N=1000, T=600 teacher-forced 10-step segments: worst err {worst[10]:.3e}")
    assert worst[10] <= 1e-3


def test_loop_headline_length_single_steps_along_the_schedule_vs_oracle(model, sd_full, dev):
    """The headline's own shape again (at T = 600 only three ten-step segments stood in for the chain): 41 SINGLE steps spread
    over the whole 1000-step schedule (every 25th step and the last one: timestep-embedding rows, coefficient rows and noise levels from t = 999 down
    to t = 0), each from latents at its own noise level against the oracle's one step from the same latents — the kernels, tile counts and the
    three-slice fused tail of the benchmarked launch at every part of the schedule (the 1 s chain above walks all 1000 steps on 60 frames)."""
    worst = _teacher_forced(model, sd_full, dev, N=1000, eta=0.0, seg_lens=[1], Ta=160000, starts={1: list(range(0, 1000, 25)) + [999]}, chain=False)
    print(f"N=1000, T=600 teacher-forced single steps at 41 places: worst err {worst[1]:.3e}")
    assert worst[1] <= 2e-4


def test_loop_997_steps_remainder_graph_teacher_forced(model, sd_full, dev):
    """Prime step count: the 997-step schedule in 57-step segments = five 10-step graphs + the 7-step remainder graph (loops below 400 steps capture ten steps per graph)."""
    worst = _teacher_forced(model, sd_full, dev, N=997, eta=0.0, seg_lens=[57], Ta=8000, starts={57: [0, 300, 640, 940]}, chain=False)
    print(f"N=997 teacher-forced 57-step segments (5 x 10 + 7): worst err {worst[57]:.3e}")
    assert worst[57] <= 1e-3


def test_loop_editing_100_steps_in_betweening(model, sd_full, dev):
    """BASELINE configs[4]'s step count (100 DDIM steps) in editing mode with the in-betweening mask, on 2 s."""
    B, Ta, N = 1, 32000, 100
    T = 120
    proc = op.process_audio([synth.synth_waveform(78, Ta).numpy()])
    init_samples = torch.sigmoid(synth.synth_latents(111, (B, T, 32))) * 0.5
    mask = torch.zeros(B, T, 32)
    mask[:, :40] = 1.0
    mask[:, 80:] = 1.0
    mask[:, :, :4] = 1.0
    en = synth.synth_latents(112, (B, T, 32))
    out = model.inference(proc.to(dev), num_inference_steps=N, guidance_scale=2.0, init_samples=init_samples.to(dev),
                          mask=mask.to(dev), edit_noise=en.to(dev))
    ref = op.inference(sd_full, proc, init_latents=en, num_inference_steps=N, guidance_scale=2.0, init_samples=init_samples,
                       mask=mask, edit_noise=en)
    err = float((out.result.cpu() - ref.result).abs().max())
    print(f"editing N=100: max abs err {err:.3e}")
    assert err <= 1e-3
    assert float((out.result.cpu()[:, :40] - init_samples[:, :40].clamp(0, 1)).abs().max()) <= 1e-6


@pytest.mark.parametrize("N", [11, 13, 23])
def test_loop_step_counts_not_divisible_by_graph_length(model, sd_full, dev, N):
    """Ten steps per graph (loops below 400 steps): N = 11, 13, 23 run N // 10 ten-step graphs + one remainder graph (free-running, whole loop); 23 also with seven steps
    per graph (said_debug_option "steps_per_graph": 3 x 7 + 2): the segmentation must not change the result."""
    eng = model._get_engine(2, 64)
    got, ref = _loop_case(model, sd_full, dev, B=1, Ta=8000, N=N, gs=2.0, tol=1e-3)
    print(f"N={N}: max abs err {float((got - ref).abs().max()):.3e}, nodes/step {model._eng.graph_num_nodes()}")
    if N == 23:
        eng.debug_option("steps_per_graph", 7)
        try:
            got2, _ = _loop_case(model, sd_full, dev, B=1, Ta=8000, N=N, gs=2.0, tol=1e-3)
        finally:
            eng.debug_option("steps_per_graph", 50)
        assert torch.equal(got, got2), "the graph segmentation must not change the result"


def test_audio_encoder_10s_vs_oracle(model, w2v_sd, dev):
    """The headline's encode: 160,000 samples -> 499 conv frames -> 600 frames."""
    proc = op.process_audio(synth.synth_waveform(8, 160000))
    ref = ow.wav2vec2_forward(w2v_sd, proc, 600)[0].numpy()
    got = model.get_audio_embedding(proc.to(dev), 600).cpu().numpy()
    err = np.abs(got - ref).max()
    print(f"audio 10 s: max abs err {err:.3e} (|ref| max {np.abs(ref).max():.2f})")
    assert err <= AUDIO_TOL * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", ["cfg", "nocfg_inter", "edit_mask_strength", "eta_rescale", "sample_pred", "v_pred_scaled", "strength0"])
def test_loop_vs_reference_own_inference_g9(golden, dev, name):
    """The HIP path against the REFERENCE's own SAID_UNet1D.inference (G9 golden, captured on CPU with diffusers stubbed
    by the oracle's scheduler: control flow, UNet and encoder are the reference's code; scheduler leg unpinned)."""
    from said_amd.model.diffusion import SAID_UNet1D
    from said_amd.model.wav2vec2 import AudioConfig
    from g9_cases import G9_CASES, g9_inputs
    c = G9_CASES[name]
    g = golden("g9_loop_control_flow_scheduler_leg_unpinned")
    m = SAID_UNet1D(audio_config=AudioConfig(num_hidden_layers=2), prediction_type=c.get("pred", "epsilon"),
                    latent_scale=c.get("latent_scale", 1))
    m.load_state_dict(synth.said_state_dict(num_w2v_layers=2), strict=True)
    m.to(dev).eval()
    proc, kw, noise, init_t = g9_inputs(c)
    kw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
    noise = {k: v.to(dev) for k, v in noise.items()}
    out = m.inference(proc.to(dev), **kw, **noise)
    ref = g[name + "_result"]
    err = float(np.abs(out.result.cpu().numpy() - ref).max())
    print(f"g9 {name}: HIP vs reference loop, max abs err {err:.2e}")
    assert err <= 1e-3
    if c.get("save_intermediate", False):
        ri = g[name + "_inter"]
        assert len(out.intermediates) == ri.shape[0]
        assert float(np.abs(torch.stack(out.intermediates).cpu().numpy() - ri).max()) <= 1e-2


def test_bf16_loop_50_steps_teacher_forced_vs_fp32_oracle(model, sd_full, dev):
    """configs[2]'s chain (50 DDIM steps, guidance) in bf16 mode against the FP32 ORACLE, teacher-forced: a free-running
    50-step chain of this random-weight network amplifies a 1e-6 perturbation 250x (oracle vs itself), so
    bf16's ~6e-3 per-evaluation rounding saturates it (measured: max 0.96, mean 0.077 — meaningless).  Each of the 50
    steps therefore starts from the oracle's latents and must land near the oracle's next latents."""
    try:
        model.set_mfma_dtype("bf16")
        worst = _teacher_forced(model, sd_full, dev, N=50, eta=0.0, seg_lens=[1], tol_single=1.0)
    finally:
        model.set_mfma_dtype("fp32")
    print(f"bf16 50-step chain, teacher-forced vs fp32 oracle: worst single-step err {worst[1]:.3e}")
    assert worst[1] <= BF16_STEP_MAX


BF16_STEP_MAX = 0.087  # 1.3x the worst measured teacher-forced bf16 step (6.7e-2: the B = 32 guided step at t = 980 of test_gpu_round3.py; the 50-step chain here: 4.3e-2)


def test_two_contexts_in_one_process(model, unet_sd, dev):
    """Two engine contexts alive at once (the timestep-frequency table used to be a process global): the
    second context is created, used and destroyed while the first keeps producing the same numbers."""
    from said_amd.model.diffusion import SAID_UNet1D
    x = synth.synth_latents(5, (2, 33, 32)).to(dev)
    c = synth.synth_latents(6, (2, 33, 768)).to(dev)
    ts = torch.tensor([77, 901]).to(dev)
    before = model(x, ts, c)
    m2 = SAID_UNet1D()
    m2.load_state_dict(synth.said_state_dict(salt=1), strict=True)
    m2.to(dev).eval()
    other = m2(x, ts, c)
    assert not torch.equal(other, before)
    m2._eng.close()
    del m2
    torch.cuda.synchronize()
    after = model(x, ts, c)
    assert torch.equal(before, after)
    ref = ou.unet1d_forward(unet_sd, x.cpu(), ts.cpu(), c.cpu())
    assert float((after.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_pred_original_sample_broadcasts_single_timestep(model, dev):
    """diffusion.py:157-186 with a length-1 / 0-dim timestep and B > 1 (reference: .view(-1, 1, 1) broadcast)."""
    B = 3
    x, n = synth.synth_latents(1, (B, 11, 32)), synth.synth_latents(2, (B, 11, 32))
    ac = model.noise_scheduler.alphas_cumprod
    for t in (torch.tensor([500]), torch.tensor(500)):
        got = model.pred_original_sample(x.to(dev), n.to(dev), t).cpu()
        a = ac[t].view(-1, 1, 1)
        want = (x - (1 - a) ** 0.5 * n) / a ** 0.5
        assert float((got - want).abs().max()) <= 1e-5


# ---------------------------------------------------------------- bf16 audio encoder (configs[2]; tgemm.hip)
BF16_AUDIO_TOL = 0.048   # 1.3x the measured 2.8e-2 (1 s) .. 3.7e-2 (10 s, 32 clips) max abs error on values up to 3.8; rms relative error measured 8e-3


@pytest.mark.parametrize("Ta,frames", [(16000, 60), (160000, 600)])
def test_bf16_audio_encoder_vs_fp32_oracle(model, w2v_sd, dev, Ta, frames):
    """bf16 mode routes the Wav2Vec2 encoder through the token-major bf16 GEMM (v_mfma_f32_32x32x16_bf16, bf16 activations
    between GEMMs, fp32 residual stream / LayerNorm / softmax).  Error against the FP32 ORACLE is reported and bounded;
    the switch must change the arithmetic and be reversible."""
    proc = op.process_audio(synth.synth_waveform(8, Ta))
    ref = ow.wav2vec2_forward(w2v_sd, proc, frames)[0]
    f32 = model.get_audio_embedding(proc.to(dev), frames).cpu()
    try:
        model.set_mfma_dtype("bf16")
        b16 = model.get_audio_embedding(proc.to(dev), frames).cpu()
    finally:
        model.set_mfma_dtype("fp32")
    again = model.get_audio_embedding(proc.to(dev), frames).cpu()
    assert torch.equal(again, f32)
    e = (b16 - ref).abs()
    rms = float((b16 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"bf16 audio encoder {Ta / 16000:g} s: max abs err {float(e.max()):.3e} (|ref| max {float(ref.abs().max()):.2f}), rms rel {rms:.3e}")
    assert b16.shape == ref.shape and torch.isfinite(b16).all()
    assert not torch.equal(b16, f32)
    assert float(e.max()) <= BF16_AUDIO_TOL and rms <= 1.2e-2


def test_bf16_unet_large_batch_token_major_gemm_path(model, unet_sd, dev):
    """bf16 mode at >= 3000 tokens per launch (5800 until round 4): the ResBlock convolutions, q/k/v, GEGLU and the folded proj_out run on the
    token-major bf16 GEMM (tgemm.hip: prep kernel + v_mfma_f32_32x32x16_bf16, channel-major fp32 results with GroupNorm
    partials).  Error against the FP32 ORACLE on the first, a middle and the last sample, and against the small-batch bf16
    path (same rounding points, different summation order)."""
    B, T = 16, 600
    x = synth.synth_latents(71, (B, T, 32))
    c = synth.synth_latents(72, (B, T, 768))
    ts = (torch.arange(B) * 61 + 5) % 1000
    try:
        model.set_mfma_dtype("bf16")
        big = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
        small = torch.cat([model(x[i:i + 1].to(dev), ts[i:i + 1].to(dev), c[i:i + 1].to(dev)).cpu() for i in (0, 7, 15)])
    finally:
        model.set_mfma_dtype("fp32")
    worst = 0.0
    for k, i in enumerate((0, 7, 15)):
        ref = ou.unet1d_forward(unet_sd, x[i:i + 1], ts[i:i + 1], c[i:i + 1])
        scale = float(ref.abs().max())
        e = float((big[i:i + 1] - ref).abs().max()) / scale
        es = float((big[i:i + 1] - small[k:k + 1]).abs().max()) / scale
        print(f"bf16 large-batch UNet sample {i}: max err {e:.2e} of range vs fp32 oracle, {es:.2e} vs the small-batch bf16 path")
        worst = max(worst, e)
    assert 1e-5 < worst <= 2e-2          # same bound as the small-batch bf16 evaluation (measured 5.9e-3 there)


def test_fp32_unet_large_batch_token_major_gemm_path(model, unet_sd, dev):
    """fp32 mode at >= 10000 tokens per launch: the ResBlock convolutions, q/k/v, GEGLU and the folded proj_out run on the
    token-major fp32 GEMM (tgemm.hip: prep kernel + fgemm_kernel on v_mfma_f32_32x32x2_f32).  Same bound as every other fp32
    UNet evaluation (1e-4 of range against the oracle); also against the small-batch channel-major path (summation order)."""
    B, T = 24, 600
    x = synth.synth_latents(73, (B, T, 32))
    c = synth.synth_latents(74, (B, T, 768))
    ts = (torch.arange(B) * 59 + 3) % 1000
    big = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    worst = 0.0
    for i in (0, 9, 23):
        small = model(x[i:i + 1].to(dev), ts[i:i + 1].to(dev), c[i:i + 1].to(dev)).cpu()
        ref = ou.unet1d_forward(unet_sd, x[i:i + 1], ts[i:i + 1], c[i:i + 1])
        scale = float(ref.abs().max())
        e = float((big[i:i + 1] - ref).abs().max()) / scale
        es = float((big[i:i + 1] - small).abs().max()) / scale
        print(f"fp32 large-batch UNet sample {i}: max err {e:.2e} of range vs oracle, {es:.2e} vs the small-batch path")
        worst = max(worst, e)
    assert worst <= 1e-4


@pytest.mark.parametrize("T", [333, 350])
def test_token_major_path_ragged_length(model, unet_sd, dev, T):
    """The token-major GEMM path (both precisions) at a sequence length that is not a multiple of the 32-token tile and whose
    two Conv1d padding rows do (T=350: 352 rows) / do not (T=333) end on a tile boundary: per-sample row pitch, padding rows,
    partial last tiles of the preparation kernel, GroupNorm partials of a short last tile."""
    B = 40                                   # 40 x 333 = 13320 tokens >= both thresholds
    x = synth.synth_latents(81, (B, T, 32))
    c = synth.synth_latents(82, (B, T, 768))
    ts = (torch.arange(B) * 53 + 11) % 1000
    pick = (0, 21, 39)
    refs = [ou.unet1d_forward(unet_sd, x[i:i + 1], ts[i:i + 1], c[i:i + 1]) for i in pick]
    out32 = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    try:
        model.set_mfma_dtype("bf16")
        out16 = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    finally:
        model.set_mfma_dtype("fp32")
    for i, ref in zip(pick, refs):
        scale = float(ref.abs().max())
        e32 = float((out32[i:i + 1] - ref).abs().max()) / scale
        e16 = float((out16[i:i + 1] - ref).abs().max()) / scale
        print(f"T={T} sample {i}: fp32 {e32:.2e}, bf16 {e16:.2e} of range")
        assert e32 <= 1e-4
        assert e16 <= 2e-2


def test_loop_large_batch_ragged_length_matches_single_clip_runs(model, sd_full, dev):
    """Guidance loop at a batch where the full-batch launches take the large-batch (token-major) kernels, at a length that is not a multiple
    of 4 or 32: every clip must come out as if run alone.  fp32: the batch against single-clip runs (same arithmetic, other tile shapes).
    bf16: the two paths round at DIFFERENT points since round 4 (large batch: bf16 activations between the kernels and bf16 attention
    operands; single clip: fp32 activations, bf16 multiplies), so each is held against the fp32 CPU ORACLE's two steps from the same
    latents, and their mutual distance is only reported."""
    B, T, N = 24, 333, 2
    ctx = synth.synth_latents(130, (B, T, 768)).to(dev)
    lat = synth.synth_latents(131, (B, T, 32)).to(dev)
    wav = torch.zeros(B, T * 16000 // 60, device=dev)   # only its shape is used when the embedding is injected
    pick = (0, 11, 23)
    refs = {i: op.inference(sd_full, wav[i:i + 1].cpu(), init_latents=lat[i:i + 1].cpu(), audio_embedding=ctx[i:i + 1].cpu(),
                            num_inference_steps=N, guidance_scale=2.0).result for i in pick}
    for mode, tol in (("fp32", 5e-5), ("bf16", BF16_LOOP2_MAX)):
        try:
            model.set_mfma_dtype(mode)
            big = model.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result
            worst = worst_big = worst_one = 0.0
            for i in pick:
                one = model.inference(wav[i:i + 1], audio_embedding=ctx[i:i + 1], num_inference_steps=N, guidance_scale=2.0,
                                      init_latents=lat[i:i + 1]).result
                assert bool(torch.isfinite(big[i]).all())
                worst = max(worst, float((big[i:i + 1] - one).abs().max()))
                worst_big = max(worst_big, float((big[i:i + 1].cpu() - refs[i]).abs().max()))
                worst_one = max(worst_one, float((one.cpu() - refs[i]).abs().max()))
        finally:
            model.set_mfma_dtype("fp32")
        print(f"{mode}: batch of {B} at T={T}: vs single-clip runs {worst:.2e}; vs oracle: batch {worst_big:.2e}, single clip {worst_one:.2e}")
        if mode == "fp32":
            assert worst <= tol and worst_big <= 1e-3 and worst_one <= 1e-3
        else:
            # (the batch against the single-clip runs is bounded too — two bf16 schedules with different rounding points, two free-running steps: measured 6.7e-2)
            assert worst_big <= tol and worst_one <= tol and worst <= tol


BF16_LOOP2_MAX = 0.09   # two free-running guided steps from pure noise (clamped result in [0, 1]); 1.35x the measured 6.6e-2 (batch) / 4.4e-2 (single clip)
