"""Round-4 GPU tests: a distribution-level statement for bf16 mode (free-running chains of this random-weight network are chaotic, so
sample-wise end-to-end bounds do not exist: DESIGN.md section 7.4 — an ENSEMBLE bound does), and alignment windows wider than the fused band
epilogue's eight keys on the large-batch shapes.  All through the C ABI."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import unet as ou  # noqa: E402
from said_amd.util import synth  # noqa: E402

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


# bounds of the ensemble statement (asserted below, quoted in DESIGN.md 7.4 / 7.6); measured values are printed by the test
ENS_MEAN_ABS = 2.8e-3     # |per-channel mean(bf16) - mean(fp32)| over 64 clips x 600 frames, result clamped to [0, 1]
ENS_STD_REL = 0.0068     # relative difference of the per-channel standard deviation
ENS_DIFF_STD_REL = 0.0095 # ... of the standard deviation of the temporal difference r[t + 1] - r[t] (the jitter an animator would see)


def test_bf16_ensemble_statistics_match_fp32_64_clips_50_steps(model, dev):
    """BASELINE configs[2]'s chain (50 DDIM steps, guidance 2) on 64 clips x 10 s, free-running, once in fp32 mode and once in bf16 mode from
    the same start noise and conditioning.  Individual samples differ (chaos amplifies bf16's rounding: printed), the ENSEMBLE does not:
    per blendshape channel, the mean and the standard deviation of the clamped result over all clips and frames, and the standard deviation
    of its temporal difference, agree within the bounds above; and the paired per-clip mean difference shows no systematic offset beyond
    four standard errors."""
    B, T, N = 64, 600, 50
    ctx = synth.synth_latents(700, (B, T, 768)).to(dev)
    lat = synth.synth_latents(701, (B, T, 32)).to(dev)
    wav = torch.zeros(B, T * 16000 // 60, device=dev)
    out = {}
    try:
        for mode in ("fp32", "bf16"):
            model.set_mfma_dtype(mode)
            out[mode] = model.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result.double().cpu()
    finally:
        model.set_mfma_dtype("fp32")
    a, b = out["fp32"], out["bf16"]
    assert torch.isfinite(b).all() and float(b.min()) >= 0 and float(b.max()) <= 1
    sample = (a - b).abs()
    m32, m16 = a.mean((0, 1)), b.mean((0, 1))
    s32, s16 = a.std((0, 1)), b.std((0, 1))
    d32, d16 = (a[:, 1:] - a[:, :-1]).std((0, 1)), (b[:, 1:] - b[:, :-1]).std((0, 1))
    clip_delta = b.mean(1) - a.mean(1)                               # (B, 32) paired per-clip differences
    se = clip_delta.std(0) / B ** 0.5
    z = (clip_delta.mean(0).abs() / se.clamp_min(1e-12))
    dm = float((m16 - m32).abs().max())
    ds = float(((s16 - s32).abs() / s32.clamp_min(1e-6)).max())
    dd = float(((d16 - d32).abs() / d32.clamp_min(1e-6)).max())
    print(f"bf16 vs fp32, 64 clips x 600 frames x 50 steps: sample-wise max |diff| {float(sample.max()):.3f} mean {float(sample.mean()):.4f}; "
          f"per-channel mean diff max {dm:.2e} (means {float(m32.min()):.3f}..{float(m32.max()):.3f}), std rel diff max {ds:.2e}, "
          f"temporal-difference std rel diff max {dd:.2e}, paired z max {float(z.max()):.2f}")
    assert dm <= ENS_MEAN_ABS and ds <= ENS_STD_REL and dd <= ENS_DIFF_STD_REL
    assert float(z.max()) <= 4.0 or dm <= 1e-3


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_wide_alignment_windows_at_large_batch(model, unet_sd, dev, mode):
    """S >> T at a batch that would take the token-major GEMM schedules (32 x 333 = 10656 tokens): windows of 9 keys — one more than the
    fused band epilogue holds — send the whole evaluation through the channel-major schedule and the generic band kernel
    (ldm/attention.py:170-189 accepts any (T, S); rounds 1-3 returned an error here)."""
    B, T, S = 32, 333, 2331        # ratio 7: windows of 9 keys
    x = synth.synth_latents(810, (B, T, 32))
    c = synth.synth_latents(811, (B, S, 768))
    ts = (torch.arange(B) * 37 + 5) % 1000
    try:
        model.set_mfma_dtype(mode)
        out = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    finally:
        model.set_mfma_dtype("fp32")
    for i in (0, 31):
        ref = ou.unet1d_forward(unet_sd, x[i:i + 1], ts[i:i + 1], c[i:i + 1])
        e = float((out[i:i + 1] - ref).abs().max()) / float(ref.abs().max())
        print(f"wide windows {mode} sample {i}: {e:.2e} of range vs oracle")
        assert e <= (1e-4 if mode == "fp32" else 2e-2)


def test_inference_encodes_identical_clips_once(model, dev):
    """The reference's batched caller passes one clip repeated (script/test_inference.py:167-168): SAID.inference encodes distinct rows once
    and gathers.  Same features up to the summation order of the encoder's GEMM tiles, which are chosen by launch size (2 clips vs 6): the
    4-step results agree to 1e-4; two of the six clips are encoded."""
    from oracle import pipeline as op
    Ta, T, N = 16000, 60, 4
    w = op.process_audio([synth.synth_waveform(900 + i, Ta).numpy() for i in range(2)]).to(dev)
    wav = w[[0, 1, 0, 0, 1, 0]].contiguous()
    lat = synth.synth_latents(901, (6, T, 32)).to(dev)
    res = {}
    for dd in (True, False):
        model.dedupe_audio = dd
        n0 = model._eng.debug_get("n_audio_clips") if model._eng is not None else 0
        res[dd] = model.inference(wav, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result
        res[(dd, "clips")] = model._eng.debug_get("n_audio_clips") - n0
    model.dedupe_audio = True
    d = float((res[True] - res[False]).abs().max())
    print(f"identical clips encoded once vs every row: max |diff| of the 4-step result {d:.2e}")
    assert d <= 1e-4
    assert res[(True, "clips")] == 2 and res[(False, "clips")] == 6


def test_device_eta_noise_50_steps_four_clips_two_groups_vs_oracle(model, dev):
    """the device Philox noise path (use_step_noise == 2) over a LONGER free-running chain at B > 1 and through clip groups
    (the generator's element counter is offset by each group's first clip): 4 clips x 0.5 s, 50 steps, eta = 1, guidance 2, two groups,
    against the CPU oracle fed said_philox_normal's values for the same seed."""
    from oracle import pipeline as op
    B, Ta, N = 4, 8000, 50
    T = int(Ta / 16000 * 60)
    proc = op.process_audio([synth.synth_waveform(950 + i, Ta).numpy() for i in range(B)])
    lat = synth.synth_latents(951, (B, T, 32))
    emb = model.get_audio_embedding(proc.to(dev), T)
    model.clip_groups = 2
    try:
        torch.manual_seed(17)
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))   # the one draw SAID.inference makes from torch's generator
        torch.manual_seed(17)
        r = model.inference(proc.to(dev), num_inference_steps=N, guidance_scale=2.0, eta=1.0, init_latents=lat.to(dev), audio_embedding=emb).result
    finally:
        model.clip_groups = None
    sn = model._eng.philox_normal(seed, 0, N, (B, T, 32)).cpu()
    ref = op.inference(synth.said_state_dict(), proc, init_latents=lat, num_inference_steps=N, guidance_scale=2.0, eta=1.0, step_noise=sn,
                       audio_embedding=emb.cpu())
    err = float((r.cpu() - ref.result).abs().max())
    print(f"device eta noise, 4 clips in two groups, 50 free-running steps: max abs err vs oracle {err:.3e}")
    assert err <= 2e-3


@pytest.mark.parametrize("gs", [1.0, 2.0])
def test_out_sched_tm_matches_channel_major_out_path(model, dev, gs):
    """The bf16 large-batch loop's last kernel on the token-major hidden state (out_sched_tm_kernel, with and without guidance, with the
    editing mask and eta noise) against the same loop with said_debug_option("out_tm", 0) — round 3's route: the last block's proj_out written
    channel-major fp32 by xgemm_kernel, then out_sched_kernel.  Same scheduler arithmetic; the model output differs by the bf16 rounding of the
    last hidden state and of out.2's weights: 2 steps stay within the bf16 per-step bound."""
    B, T, N = 24, 600, 2
    ctx = synth.synth_latents(960, (B, T, 768)).to(dev)
    init = synth.synth_latents(961, (B, T, 32)).abs().clamp(0, 1).to(dev)
    en = synth.synth_latents(962, (B, T, 32)).to(dev)
    mask = torch.zeros(B, T, 32, device=dev)
    mask[:, 100:300] = 1
    wav = torch.zeros(B, T * 16000 // 60, device=dev)
    res = {}
    try:
        model.set_mfma_dtype("bf16")
        for ot in (0, 1):
            model._get_engine(2 * B, T).debug_option("out_tm", ot)
            torch.manual_seed(5)
            res[ot] = model.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=gs, eta=1.0, init_samples=init, mask=mask,
                                      edit_noise=en).result
        nodes = model._eng.graph_num_nodes()
    finally:
        model._eng.debug_option("out_tm", -1)
        model.set_mfma_dtype("fp32")
    d = float((res[0] - res[1]).abs().max())
    print(f"out_sched_tm vs channel-major out path, guidance {gs}: max |diff| after {N} steps {d:.3e} ({nodes} graph nodes per step)")
    assert torch.isfinite(res[1]).all() and d <= 0.087
    # the masked frames are add_noise(init_samples) at the next timestep whatever the model output: bit-identical between the two routes
    # (the kernel's arithmetic itself is checked against the stand-alone scheduler in tests/test_gpu_round5.py)
    assert torch.equal(res[1][:, 100:300], res[0][:, 100:300])


# ---------------------------------------------------------------- split-fp16 attention products (fp32 mode)
def _f64_truth(unet_sd, x, ts, c):
    """The oracle's op sequence evaluated in float64 on the same fp32 inputs and weights (its `.float()` casts redirected): the yardstick both
    product modes are measured against — the fp32 oracle itself sits 1e-6 of the output range away from it."""
    from unittest import mock
    sd64 = {k: v.double() for k, v in unet_sd.items()}
    with mock.patch.object(torch.Tensor, "float", torch.Tensor.double):
        return ou.unet1d_forward(sd64, x.double(), ts, c.double())


@pytest.mark.parametrize("B,T", [(1, 600), (2, 37), (12, 600), (20, 600)])
def test_split_fp16_attention_is_as_close_to_float64_as_fp32_mfma(model, unet_sd, dev, B, T):
    """Split-fp16 products (split_f16.h: x = h + 2^-11 l, three fp16 MFMAs per fp32 one, fp32 accumulation): the large-batch fp32 GEMMs by default
    (fgemm_kernel SP), the two self-attention products as an option (attn.hip, PM == 2).  Statement: against the float64 evaluation of the same network its UNet output is no further away than the
    v_mfma_f32_32x32x2_f32 path's (said_debug_option("attn_split", 0)) beyond a factor 1.5, and both stay inside 2e-5 of the output range
    (the stated single-evaluation tolerance is 1e-4).  (12, 600) takes the large-batch attention (four query tiles per workgroup); (20, 600) — 12000 UNet rows —
    the large-batch fp32 schedule, whose token-major GEMMs (fgemm_kernel) run on split-fp16 operands as well (said_debug_option("gemm_split", 0 / 1)): both
    options are switched together."""
    x = synth.synth_latents(700 + B, (B, T, 32))
    c = synth.synth_latents(800 + B, (B, T, 768))
    ts = (torch.arange(B) * 83 + 999) % 1000
    nref = min(B, 2)
    truth = _f64_truth(unet_sd, x[:nref], ts[:nref], c[:nref])
    scale = float(truth.abs().max())
    err = {}
    try:
        for sp in (0, 1):
            model._get_engine(2 * B, T).debug_option("attn_split", sp)
            model._eng.debug_option("gemm_split", sp)
            out = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
            err[sp] = float((out[:nref].double() - truth).abs().max()) / scale
    finally:
        model._eng.debug_option("attn_split", -1)
        model._eng.debug_option("gemm_split", -1)
    print(f"\n[attn_split] B={B} T={T}: fp32 MFMA {err[0]:.3e}, split fp16 {err[1]:.3e} of the output range")
    assert err[0] <= 2e-5 and err[1] <= 2e-5
    assert err[1] <= 1.5 * err[0] + 1e-6


def test_split_fp16_attention_small_magnitudes(model, unet_sd, dev):
    """Operands far below fp16's normal range: the same network with every to_v weight of the self-attention scaled by 2^-14 (v ~ 1e-5:
    h is an fp16 denormal or 0, l a denormal) and to_out scaled back by 2^14 — exact powers of two, so the fp32 result is unchanged up to
    underflow.  A matrix pipe that flushed fp16 denormals would lose v here."""
    sd = {k: v.clone() for k, v in unet_sd.items()}
    for k in sd:
        if ".attn1.to_v.weight" in k:
            sd[k] *= 2.0 ** -14
        if ".attn1.to_out.0.weight" in k:
            sd[k] *= 2.0 ** 14
    from said_amd.model.diffusion import SAID_UNet1D
    m = SAID_UNet1D()
    full = synth.said_state_dict()
    for k, v in sd.items():
        full["denoiser." + k] = v
    m.load_state_dict(full, strict=True)
    m.to(dev).eval()
    m._get_engine(2, 96).debug_option("attn_split", 1)   # (opt-in since the bit-stable issue order turned out no faster than the fp32 MFMAs)
    x = synth.synth_latents(901, (1, 96, 32))
    c = synth.synth_latents(902, (1, 96, 768))
    ts = torch.tensor([500])
    truth = _f64_truth(sd, x, ts, c)
    out = m(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    err = float((out.double() - truth).abs().max()) / float(truth.abs().max())
    print(f"\n[attn_split] v scaled by 2^-14: {err:.3e} of the output range")
    assert err <= 5e-5


@pytest.mark.parametrize("dt,groups,opts", [("fp32", 3, {}), ("fp32", 3, {"attn_split": 1}), ("fp32", 1, {"gemm_split": 1, "attn_split": 1}), ("bf16", 3, {})])
def test_split_fp16_attention_is_deterministic_under_concurrent_clip_groups(dev, dt, groups, opts):
    """Regression guard for profiles/r04i_attn_split_hazard.txt: 32 clips, four repetitions, bit for bit — as THREE concurrent clip groups with the defaults
    (fp32 MFMAs / bf16) and with the opt-in split-fp16 attention (three accumulators in rotation: the issue order that is bit-stable next to other streams; the
    faster orders differed in a few clips by up to 5e-2 in most repetitions), and as ONE group with the opt-in split-fp16 GEMMs (with three groups one soak
    process in twelve deviated: that combination is documented as not bit-stable and is not asserted here)."""
    from said_amd.model.diffusion import SAID_UNet1D
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    m.set_mfma_dtype(dt)
    m.clip_groups = groups
    B, T = 32, 600
    for k, v in opts.items():   # clones copy the options when they are created (first inference)
        m._get_engine(2 * B, T).debug_option(k, v)
    ctx = synth.synth_latents(700 + B, (B, T, 768)).to(dev)
    lat = synth.synth_latents(800 + B, (B, T, 32)).to(dev)
    wav = torch.zeros(B, T * 16000 // 60, device=dev)
    first = None
    for rep in range(4):
        r = m.inference(wav, audio_embedding=ctx, num_inference_steps=2, guidance_scale=2.0, init_latents=lat).result
        assert torch.isfinite(r).all()
        if first is None:
            first = r.clone()
        else:
            assert torch.equal(r, first), f"repetition {rep}: max abs diff {float((r - first).abs().max())}"
    m._eng.close()
