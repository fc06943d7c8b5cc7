"""Round 5 (run on the MI355X with `-m gpu`): the fp32 mode's new arithmetic and schedule.

* small-batch GEMMs on split-fp16 operands (gemm_lds.hip SP; said_debug_option "ugemm_split");
* the per-token tail of a SpatialTransformer as one launch (stchain.hip; "st_chain"): against the five-launch schedule it replaces
  and against the CPU oracle, at ragged lengths, under guidance with odd clip counts (shared prefix), with its fall-backs;
* the split's consistency (split_f16.h): the high and low planes of an operand must come from ONE fp32 -> fp16 conversion — the
  regression that showed as single tokens off by 1e-4 (an operand off by a whole fp16 ulp wherever a value sat exactly half-way
  between two fp16 numbers).
"""
import numpy as np
import pytest
import torch

from oracle import pipeline as op
from oracle import scheduler as osch
from oracle import unet as ou
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
def sd_parts():
    sd = synth.said_state_dict()
    return (sd,) + tuple(op.split_state_dict(sd))   # full, audio, unet, null embedding


def _forward(model, dev, B, T, S=None, opts=None, seed=1):
    S = T if S is None else S
    x = synth.synth_latents(seed, (B, T, 32))
    c = synth.synth_latents(seed + 1, (B, S, 768))
    ts = (torch.arange(B) * 137 + 500) % 1000
    eng = model._get_engine(max(B, 2), max(T, 64))
    for k, v in (opts or {}).items():
        eng.debug_option(k, v)
    try:
        n0 = eng.debug_get("n_stchain")
        out = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
        n1 = eng.debug_get("n_stchain")
    finally:
        for k in (opts or {}):
            eng.debug_option(k, -1 if k in ("st_chain", "ugemm_split") else 0)
    return out, (x, ts, c), n1 - n0


# measured (MI355X, round 5): chain vs five launches <= 6e-7 of the output range; before the one-conversion fix the same comparison gave
# 4.8e-6 .. 1.3e-5 (single tokens off by 1e-4 after the first transformer block)
CHAIN_VS_FIVE = 1e-6


@pytest.mark.parametrize("B,T", [(2, 600), (1, 37), (3, 333), (2, 60)])
def test_stchain_matches_the_five_launch_schedule(model, dev, B, T):
    a, _, na = _forward(model, dev, B, T, opts={"st_chain": 1})
    b, _, nb = _forward(model, dev, B, T, opts={"st_chain": 0})
    assert na == 4 and nb == 0, "one fused launch per transformer block / none with the option off"
    rng = float(b.max() - b.min())
    err = float((a - b).abs().max()) / rng
    print(f"stchain vs five launches B={B} T={T}: {err:.2e} of range")
    assert err <= CHAIN_VS_FIVE


@pytest.mark.parametrize("B,T", [(2, 60), (1, 600)])
def test_stchain_and_split_gemms_vs_oracle(model, dev, sd_parts, B, T):
    """Each combination of the two options against the CPU oracle at the fp32 bound of the parity suite (1e-4 of range; measured 4e-7 .. 6e-7)."""
    _, _, sd_u, _ = sd_parts
    ref = None
    for opts in ({"st_chain": 1, "ugemm_split": 1}, {"st_chain": 0, "ugemm_split": 1}, {"st_chain": 0, "ugemm_split": 0}, {"st_chain": 1, "ugemm_split": 0}):
        out, (x, ts, c), _ = _forward(model, dev, B, T, opts=opts)
        if ref is None:
            ref = ou.unet1d_forward(sd_u, x, ts, c)
        err = float((out - ref).abs().max()) / float(ref.max() - ref.min())
        print(f"B={B} T={T} {opts}: {err:.2e} of range")
        assert err <= 1e-4


def test_stchain_falls_back_when_the_window_tile_does_not_fit(model, dev, sd_parts):
    """S >> T: a 32-token tile's alignment windows span more than the kernel's 56 key rows (or are wider than 8 keys): the five launches run."""
    _, _, sd_u, _ = sd_parts
    out, (x, ts, c), n = _forward(model, dev, 2, 40, S=400, seed=7)
    assert n == 0
    ref = ou.unet1d_forward(sd_u, x, ts, c)
    assert float((out - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # S < T (windows advance slower than the tokens) and S slightly above T fit
    for S in (25, 50):
        out, (x, ts, c), n = _forward(model, dev, 1, 40, S=S, seed=9)
        assert n == 4
        ref = ou.unet1d_forward(sd_u, x, ts, c)
        assert float((out - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("B,Ta", [(1, 16000), (3, 9867), (2, 160000)])
def test_guided_step_on_the_fused_schedule_vs_oracle(model, dev, sd_parts, B, Ta):
    """One guided DDIM step (shared prefix in the first block: one workgroup row per clip feeds both halves; unconditional samples skip the
    cross-attention) from the same latents as the oracle: odd clip counts, a ragged length (616 frames... 9867 samples = 37 frames), 10 s clips."""
    sd_full, sd_a, sd_u, null = sd_parts
    T = int(Ta / 16000 * 60)
    proc = op.process_audio([synth.synth_waveform(10 + i, Ta).numpy() for i in range(B)])
    emb = op.get_audio_embedding(sd_a, proc, T)
    lat = synth.synth_latents(100, (B, T, 32))
    o = osch.OracleDDIM()
    o.set_timesteps(50)
    sch = model.noise_scheduler
    sch.set_timesteps(50)
    ts = sch.timesteps.numpy()
    coef = sch.coef_table(ts, 0.0)
    k = 24
    tt = int(ts[k])
    pred = ou.unet1d_forward(sd_u, torch.cat([lat] * 2), torch.tensor([tt] * (2 * B)), torch.cat([null.repeat(B, T, 1), emb]))
    e_u, e_c = pred.chunk(2)
    want = o.step(e_c + 2.0 * (e_c - e_u), tt, lat)
    eng = model._get_engine(2 * B, T)
    n0 = eng.debug_get("n_stchain")
    _, latf, _ = eng.denoise_loop(latents=lat.to(dev), context=emb.to(dev), timesteps=ts[k:k + 1], coef=coef[k:k + 1], prediction_type="epsilon",
                                  guidance_scale=2.0, guidance_rescale=0.0, latent_scale=1.0, step_noise=None)
    assert eng.debug_get("n_stchain") > n0
    err = float((latf.cpu() - want).abs().max())
    print(f"guided step B={B} T={T}: max abs err {err:.2e}")
    assert err <= 2e-4   # (measured 4e-6 .. 6e-6; the parity suite's single-step bound)


def test_out_sched_split_products_against_fp32_products(model, dev, sd_parts):
    """out_sched_kernel's convolution on split-fp16 operands (the default) against the fp32 matrix instructions (said_debug_option "out_split" = 0) over five guided
    steps from the same latents: same scheduler arithmetic, the model output differs by the products' rounding only."""
    sd_full, sd_a, sd_u, null = sd_parts
    B, T = 2, 190
    lat = synth.synth_latents(77, (B, T, 32))
    emb = synth.synth_latents(78, (B, T, 768))
    sch = model.noise_scheduler
    sch.set_timesteps(50)
    ts = sch.timesteps.numpy()
    coef = sch.coef_table(ts, 0.0)
    eng = model._get_engine(2 * B, T)
    res = {}
    for v in (1, 0):
        eng.debug_option("out_split", v)
        try:
            _, latf, _ = eng.denoise_loop(latents=lat.to(dev), context=emb.to(dev), timesteps=ts[20:25], coef=coef[20:25], prediction_type="epsilon",
                                          guidance_scale=2.0, guidance_rescale=0.0, latent_scale=1.0, step_noise=None)
            res[v] = latf.cpu()
        finally:
            eng.debug_option("out_split", -1)
    d = float((res[1] - res[0]).abs().max())
    print(f"out_sched split vs fp32 products, 5 guided steps: max |diff| {d:.2e} (latent range {float(res[0].abs().max()):.2f})")
    assert torch.isfinite(res[1]).all() and d <= 2e-5


@pytest.mark.parametrize("B,stores", [(2, False), (16, True)])
def test_bf16_mode_against_the_bf16_emulating_oracle(model, dev, sd_parts, B, stores):
    """VERDICT r4 #3b, end to end — and why the kernel-by-kernel test below is the instrument.  bf16 mode's whole UNet evaluation beside the oracle with the SAME
    roundings switched on (oracle/unet.py ROUND_OPERANDS; ROUND_STORES for the large-batch schedule, whose hidden state is stored in bf16 between kernels).  Measured:
    the deviation from the emulation (rms 1.1e-3 of range at B = 2) is hardly smaller than the deviation from the fp32 oracle (1.35e-3), although every kernel agrees
    with the emulation of its OWN inputs to 1e-7 .. 1e-5.  A rounding is a discontinuity: a 1e-6 difference in an operand that sits next to a bf16 boundary becomes a
    2^-9 relative one, those differences create more flips downstream, and within one SpatialTransformer two bf16 evaluations with different summation orders have
    decorrelated to the size of the rounding noise itself (scripts/probe_bf16_emu.py prints the growth launch by launch: 5e-6, 1.3e-5, 6e-5, 1.6e-4, 3e-4 of range).
    So end to end only the bound against the fp32 oracle is asserted; the figures against the emulation are printed for the record."""
    sd_full, sd_a, sd_u, null = sd_parts
    T = 600
    x = synth.synth_latents(171, (B, T, 32))
    c = synth.synth_latents(172, (B, T, 768))
    ts = (torch.arange(B) * 61 + 5) % 1000
    eng = model._get_engine(max(B, 2), T)
    try:
        model.set_mfma_dtype("bf16")
        n0 = eng.debug_get("n_stchain")
        out = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
        chained = eng.debug_get("n_stchain") > n0
    finally:
        model.set_mfma_dtype("fp32")
    assert chained == stores, "the large batch runs the token-major bf16 schedule with the fused tail, the small one does not"
    worst = (0.0, 0.0, 0.0, 0.0)
    for i in sorted({0, B // 2, B - 1}):
        ref = ou.unet1d_forward(sd_u, x[i:i + 1], ts[i:i + 1], c[i:i + 1])
        try:
            ou.ROUND_OPERANDS, ou.ROUND_STORES, ou.ATTN_KS = "bf16", stores, (None if stores else 4)   # (battn_kernel's tiling is not restated: plain softmax there)
            emu = ou.unet1d_forward(sd_u, x[i:i + 1], ts[i:i + 1], c[i:i + 1])
        finally:
            ou.ROUND_OPERANDS, ou.ROUND_STORES, ou.ATTN_KS = None, False, 4
        rng = float(ref.abs().max())
        d_emu, d_ref = out[i:i + 1] - emu, out[i:i + 1] - ref
        m_emu, m_ref = float(d_emu.abs().max()) / rng, float(d_ref.abs().max()) / rng
        r_emu, r_ref = float(d_emu.pow(2).mean().sqrt()) / rng, float(d_ref.pow(2).mean().sqrt()) / rng
        print(f"bf16 mode B={B} sample {i}: vs emulating oracle max {m_emu:.2e} rms {r_emu:.2e} of range | vs fp32 oracle max {m_ref:.2e} rms {r_ref:.2e}")
        worst = max(worst, (m_emu, r_emu, m_ref, r_ref))
        assert m_ref <= 2e-2
    print(f"worst: {worst}")


def test_bf16_kernels_against_the_operand_rounded_evaluation_of_their_own_inputs(model, dev, sd_parts):
    """VERDICT r4 #3b, kernel by kernel.  bf16 mode, small-batch schedule (B = 2, T = 600): the schedule is stopped behind each of the first ten launches (first
    ResBlock, first SpatialTransformer), the launch's INPUT buffers are read back, and the stage is re-evaluated on the CPU with both operands of every product rounded
    to bf16 (float64 accumulation; oracle/unet.py ROUND_OPERANDS) — statistics, normalisations, activations, biases and residual sums in fp32 as in the kernels.  On its own
    inputs every GEMM-type kernel must agree with that evaluation at the level of single rounding flips, 2-3 orders of magnitude below its distance from the fp32
    evaluation of the same inputs (which is the bf16 rounding itself); self-attention is restated with the kernel's online softmax (it rounds probabilities relative
    to a running maximum per key slice)."""
    import torch.nn.functional as F
    sd_full, sd_a, sd = sd_parts[0], sd_parts[1], sd_parts[2]
    B, T = 2, 600
    Tp = (T + 31) // 32 * 32
    x = synth.synth_latents(181, (B, T, 32))
    c = synth.synth_latents(182, (B, T, 768))
    ts = torch.tensor([999, 17])
    eng = model._get_engine(2, T)
    bufs = {}

    def grab(k, name, rows, C):
        eng.debug_stop_after(k)
        model(x.to(dev), ts.to(dev), c.to(dev))
        bufs[(k, name)] = torch.from_numpy(eng.debug_read(name, (B, rows, Tp))[:, :C, :T].copy())

    try:
        model.set_mfma_dtype("bf16")
        for k, name, rows, C in ((1, "H0", 192, 192), (2, "M", 192, 192), (3, "P", 192, 192), (5, "O", 384, 192), (6, "X1", 192, 192), (7, "O", 384, 192), (8, "X2", 192, 192),
                                 (9, "F", 768, 768), (10, "H1", 192, 192)):
            grab(k, name, rows, C)
    finally:
        eng.debug_stop_after(-1)
        model.set_mfma_dtype("fp32")
    H0, M, P, O1, X1, O2, X2, Fg, H1 = (bufs[k] for k in ((1, "H0"), (2, "M"), (3, "P"), (5, "O"), (6, "X1"), (7, "O"), (8, "X2"), (9, "F"), (10, "H1")))
    rb, st = "model.input_blocks.1.0", "model.input_blocks.1.1"
    tb = st + ".transformer_blocks.0"
    emb = ou.time_embed(sd, ts)
    e = F.linear(F.silu(emb), sd[rb + ".emb_layers.1.weight"], sd[rb + ".emb_layers.1.bias"])
    ln = lambda t, n: F.layer_norm(t, (192,), sd[tb + f".norm{n}.weight"], sd[tb + f".norm{n}.bias"])

    def stages(rounded):
        out = {}
        ou.ROUND_OPERANDS = "bf16" if rounded else None
        try:
            h = F.silu(F.group_norm(H0, 32, sd[rb + ".in_layers.0.weight"], sd[rb + ".in_layers.0.bias"], eps=1e-5))
            out["2 ResBlock conv 1"] = (ou._conv1d(h, sd[rb + ".in_layers.2.weight"], None, padding=1) + (sd[rb + ".in_layers.2.bias"][None] + e)[..., None], M)
            h = F.silu(F.group_norm(M, 32, sd[rb + ".out_layers.0.weight"], sd[rb + ".out_layers.0.bias"], eps=1e-5))
            out["3 ResBlock conv 2 + x"] = (ou._conv1d(h, sd[rb + ".out_layers.3.weight"], sd[rb + ".out_layers.3.bias"], padding=1) + H0, P)
            g = F.group_norm(P, 32, sd[st + ".norm.weight"], sd[st + ".norm.bias"], eps=1e-6).transpose(1, 2)
            y = ln(g, 1)
            q, k, v = (ou._linear(y, sd[tb + f".attn1.to_{n}.weight"]) for n in "qkv")
            sp = lambda t: t.reshape(B, T, 6, 32).permute(0, 2, 1, 3).reshape(B * 6, T, 32)
            if rounded:
                a = ou.attn_bf16_online(sp(q), sp(k), sp(v), 4)
            else:
                a = torch.einsum("bij,bjd->bid", (torch.einsum("bid,bjd->bij", sp(q), sp(k)) * 32 ** -0.5).softmax(dim=-1), sp(v))
            out["4+5 q/k/v + self-attention"] = (a.reshape(B, 6, T, 32).permute(0, 2, 1, 3).reshape(B, T, 192).transpose(1, 2), O1)
            out["6 to_out + norm(x)"] = ((ou._linear(O1.transpose(1, 2), sd[tb + ".attn1.to_out.0.weight"], sd[tb + ".attn1.to_out.0.bias"]) + g).transpose(1, 2), X1)
            x1 = X1.transpose(1, 2)
            q2 = ou._linear(ln(x1, 2), sd[tb + ".attn2.to_q.weight"])
            k2, v2 = F.linear(c, sd[tb + ".attn2.to_k.weight"]), F.linear(c, sd[tb + ".attn2.to_v.weight"])
            mask = ou.alignment_mask(B, T, T)
            sim = (torch.einsum("bid,bjd->bij", sp(q2), sp(k2)) * 32 ** -0.5).masked_fill(mask[:, None].expand(B, 6, T, T).reshape(B * 6, T, T), -torch.finfo(torch.float32).max)
            a2 = torch.einsum("bij,bjd->bid", sim.softmax(dim=-1), sp(v2))
            out["7 to_q + banded cross-attention"] = (a2.reshape(B, 6, T, 32).permute(0, 2, 1, 3).reshape(B, T, 192).transpose(1, 2), O2)
            out["8 to_out + x1"] = ((ou._linear(O2.transpose(1, 2), sd[tb + ".attn2.to_out.0.weight"], sd[tb + ".attn2.to_out.0.bias"]) + x1).transpose(1, 2), X2)
            x2 = X2.transpose(1, 2)
            yv = ou._linear(ln(x2, 3), sd[tb + ".ff.net.0.proj.weight"], sd[tb + ".ff.net.0.proj.bias"])
            av, gate = yv.chunk(2, dim=-1)
            out["9 GEGLU"] = ((av * F.gelu(gate)).transpose(1, 2), Fg)
            Pw = sd[st + ".proj_out.weight"].reshape(192, 192).double()
            PF = (Pw @ sd[tb + ".ff.net.2.weight"].double()).float()
            PB = (Pw @ sd[tb + ".ff.net.2.bias"].double() + sd[st + ".proj_out.bias"].double()).float()
            out["10 folded proj_out + x_in"] = ((ou._linear(Fg.transpose(1, 2), PF) + ou._linear(x2, Pw.float()) + PB).transpose(1, 2) + P, H1)
        finally:
            ou.ROUND_OPERANDS = None
        return out

    emu, plain = stages(True), stages(False)
    for name in emu:
        want, got = emu[name]
        rng = float(got.abs().max())
        d = (got - want).abs()
        d32 = (got - plain[name][0]).abs()
        mx, rms, rms32 = float(d.max()) / rng, float(d.pow(2).mean().sqrt()) / rng, float(d32.pow(2).mean().sqrt()) / rng
        print(f"bf16 kernel vs its inputs' operand-rounded evaluation | {name:34s}: max {mx:.2e} rms {rms:.2e} of range"
              f"   (vs the unrounded evaluation: max {float(d32.max()) / rng:.2e} rms {rms32:.2e})")
        # measured (B = 2, T = 600): products + residual (6, 8, 10) max 6e-8 .. 1.5e-7; convolutions behind GroupNorm + SiLU rms 8e-7 .. 2.8e-6, max 1.7e-4 (an operand next to
        # a bf16 boundary rounds the other way: one flip is 2^-9 of that operand); self-attention rms 1.2e-5, max 6e-4; cross-attention 4.6e-6; GEGLU 4e-7 — against
        # 1.1e-4 .. 6.2e-4 rms for the unrounded evaluation of the same inputs
        if name.startswith(("6 ", "8 ", "10 ")):
            assert mx <= 1e-6
        else:
            assert rms <= 4e-5 and rms <= 0.1 * rms32 and mx <= 2e-3


def test_stchain_bf16_beside_the_six_launch_tail(model, dev, sd_parts):
    """The token-major bf16 schedule with the fused tail (stchain_kernel<bf16>, two workgroups per CU) and with rgemm's six launches (said_debug_option
    "st_chain_bf16" = 0): two bf16 evaluations with different rounding points (the fused kernel keeps x1 / x2 in fp32 registers and rounds them only as operands; its
    K / V window tiles are bf16) — each within the bf16 bound of the fp32 oracle, equally far from it, and no further from each other than from the oracle."""
    sd_u = sd_parts[2]
    B, T = 16, 600
    x = synth.synth_latents(191, (B, T, 32))
    c = synth.synth_latents(192, (B, T, 768))
    ts = (torch.arange(B) * 53 + 11) % 1000
    eng = model._get_engine(B, T)
    res, used = {}, {}
    try:
        model.set_mfma_dtype("bf16")
        for v in (1, 0):   # one tile per workgroup (stchain_kernel<bf16>, the default), six launches (round 5's two-tile variant was removed in round 6: measured slower)
            eng.debug_option("st_chain_bf16", v)
            n0 = eng.debug_get("n_stchain")
            res[v] = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
            used[v] = eng.debug_get("n_stchain") - n0
    finally:
        eng.debug_option("st_chain_bf16", -1)
        model.set_mfma_dtype("fp32")
    assert used[1] == 3 and used[0] == 0   # (the last block's tail feeds out_sched_tm's input through the rgemm pair)
    rms = lambda d, rng: float(d.pow(2).mean().sqrt()) / rng
    for i in (0, B - 1):
        ref = ou.unet1d_forward(sd_u, x[i:i + 1], ts[i:i + 1], c[i:i + 1])
        rng = float(ref.abs().max())
        e1, e0, e10 = rms(res[1][i:i + 1] - ref, rng), rms(res[0][i:i + 1] - ref, rng), rms(res[1][i:i + 1] - res[0][i:i + 1], rng)
        m1 = float((res[1][i:i + 1] - ref).abs().max()) / rng
        print(f"bf16 B={B} sample {i}: fused tail vs fp32 oracle rms {e1:.2e} (max {m1:.2e}), six launches {e0:.2e}, fused vs six launches {e10:.2e} of range")
        assert m1 <= 2e-2 and e1 <= 1.3 * e0 + 1e-4 and e10 <= 1.5 * max(e1, e0)


@pytest.mark.parametrize("B,T", [(2, 600), (1, 37), (3, 333), (2, 1800)])
def test_presplit_kv_attention_matches_splitting_in_the_attention_kernel(model, dev, sd_parts, B, T):
    """fp32 small batch: the q/k/v GEMM stores k and v as packed split-fp16 pairs (one split per element) and attn_kernel<PM = 3> unpacks them, instead of every
    query-tile workgroup splitting all of K and V again (said_debug_option "attn_presplit" = 0).  The same h / l planes reach the same MFMAs — up to the one
    element in 8192 where the scalar conversion of the GEMM's epilogue and the packed conversion of the attention kernel round a tie differently — so the two
    agree far below the oracle bound; both against the oracle."""
    sd_u = sd_parts[2]
    x = synth.synth_latents(221, (B, T, 32))
    c = synth.synth_latents(222, (B, T, 768))
    ts = (torch.arange(B) * 331 + 7) % 1000
    eng = model._get_engine(max(B, 2), T)
    res = {}
    for v in (1, 0):
        eng.debug_option("attn_presplit", v)
        try:
            res[v] = model(x.to(dev), ts.to(dev), c.to(dev)).cpu()
        finally:
            eng.debug_option("attn_presplit", -1)
    ref = ou.unet1d_forward(sd_u, x, ts, c)
    rng = float(ref.abs().max())
    d10 = float((res[1] - res[0]).abs().max()) / rng
    e1, e0 = float((res[1] - ref).abs().max()) / rng, float((res[0] - ref).abs().max()) / rng
    print(f"pre-split k / v B={B} T={T}: vs splitting in the kernel {d10:.2e} of range; vs oracle {e1:.2e} (off: {e0:.2e})")
    assert d10 <= 2e-6 and e1 <= 1e-4 and e0 <= 1e-4


def test_split_planes_come_from_one_conversion(model, dev):
    """The regression itself, at the API: latents / context chosen so that many LayerNorm / attention outputs cannot be known in advance — instead the
    property is checked where it bites: the fused schedule (packed conversions in its epilogues) against the five-launch one, PER TOKEN.  With the
    planes taken from two conversions, single tokens of the first block's output were off by 1e-4 (8e-6 of range) while the rest agreed to 1e-6."""
    B, T = 2, 600
    Tp = (T + 31) // 32 * 32
    eng = model._get_engine(max(B, 2), T)
    x = synth.synth_latents(1, (B, T, 32)).to(dev)
    c = synth.synth_latents(2, (B, T, 768)).to(dev)
    ts = torch.tensor([500] * B).to(dev)
    res = {}
    for name, ch, stop in (("chain", 1, 6), ("five", 0, 10)):   # launches up to the end of the first transformer block
        eng.debug_option("st_chain", ch)
        eng.debug_stop_after(stop)
        try:
            model(x, ts, c)
            res[name] = eng.debug_read("H1", (B, 192, Tp))[:, :, :T].copy()
        finally:
            eng.debug_stop_after(-1)
            eng.debug_option("st_chain", -1)
    d = np.abs(res["chain"] - res["five"]).max(axis=(0, 1))   # per token
    rng = float(res["five"].max() - res["five"].min())
    print(f"first block, per-token max |chain - five|: median {np.median(d):.2e}, max {d.max():.2e} (range {rng:.1f})")
    assert d.max() <= 1.5e-5, "a token is off by far more than rounding: split planes from two different conversions?"


def test_show_process_twice_does_not_start_from_the_previous_count(model, dev):
    """the progress poller used to be able to read the PREVIOUS loop's final step count before the new loop's reset had executed (the bar
    jumped to `total` at once).  The wrapper now resets the counter synchronously (said_loop_progress_reset) before it starts polling."""
    from said_amd.model import diffusion as D
    proc = op.process_audio(synth.synth_waveform(3, 16000)).to(dev)
    lat = synth.synth_latents(9, (1, 60, 32)).to(dev)
    seen = []
    orig = D._Progress._advance

    def spy(self):
        orig(self)
        seen[-1].append(self._done)

    D._Progress._advance = spy
    try:
        for _ in range(2):
            seen.append([])
            model.inference(proc, num_inference_steps=40, guidance_scale=2.0, init_latents=lat, show_process=True)
            eng = model._eng
            assert eng.loop_progress() == 40
        eng.loop_progress_reset()
        assert eng.loop_progress() == 0
    finally:
        D._Progress._advance = orig
    for s in seen:
        assert s and s == sorted(s) and s[-1] == 40


# ---------------------------------------------------------------- bf16 large batch: the step's last kernel against the tested scheduler (VERDICT r4 #3a)
def _gn_coef_from_partials(st, T, gamma, beta, eps):
    """GroupNorm(32 groups of 6 channels) coefficients (a, b) per (sample, channel) from a producer's partial statistics [sample][32-token tile][192][(mean, M2)]:
    what the consuming kernels finalise (gemm_common.h gn_finish) — the statistics of the producer's fp32 values, not of their bf16 store."""
    Be, npart = st.shape[0], st.shape[1]
    cnt = torch.tensor([min(32, T - 32 * p) for p in range(npart)], dtype=torch.float64).view(1, npart, 1)
    mean_pc, m2_pc = st[..., 0].double(), st[..., 1].double()
    gm = (cnt * mean_pc).view(Be, npart, 32, 6).sum(dim=(1, 3)) / (6 * T)
    gm_c = gm.repeat_interleave(6, dim=1).view(Be, 1, 192)
    m2 = (m2_pc + cnt * (mean_pc - gm_c) ** 2).view(Be, npart, 32, 6).sum(dim=(1, 3))
    rstd = 1.0 / torch.sqrt(m2 / (6 * T) + eps)
    a = (gamma.double().view(1, 192) * rstd.repeat_interleave(6, dim=1)).float()
    b = (beta.double().view(1, 192) - gm.repeat_interleave(6, dim=1) * a.double()).float()
    return a, b   # (Be, 192) each


def _ws(eng, name, nbytes_max=None):
    for i, nm, nb in eng.ws_buffers():
        if nm == name:
            t = eng.ws_snapshot(i, nb if nbytes_max is None else min(nb, nbytes_max))
            torch.cuda.synchronize()
            return t
    raise KeyError(name)


def test_stchain_bf16_against_the_operand_rounded_evaluation_of_its_own_inputs(model, dev, sd_parts):
    """VERDICT r4 #3b for the kernel this round added to configs[2]'s schedule.  Token-major bf16 schedule (16 samples x 600 frames): the evaluation is stopped behind the
    first stchain_kernel<bf16> launch, its inputs are read back as stored (attention output and block input in bf16, the context), and the whole tail — to_out + GroupNorm'ed
    residual, LayerNorm, to_q, banded cross-attention over bf16 K / V tiles, to_out, LayerNorm, GEGLU, folded proj_out + x_in — is re-evaluated on the CPU with the
    kernel's operand roundings and ITS weights (LayerNorm affines folded into the next product's weights before rounding, proj_out o ff.net.2 formed in double:
    engine.cpp pack_chain), float64 accumulation, everything else fp32.  The stored bf16 result must be the rounding of that evaluation almost everywhere."""
    import torch.nn.functional as F
    sd = sd_parts[2]
    B, T = 16, 600
    seg = (T + 63) // 64 * 64
    x = synth.synth_latents(201, (B, T, 32))
    c = synth.synth_latents(202, (B, T, 768))
    ts = (torch.arange(B) * 47 + 3) % 1000
    eng = model._get_engine(B, T)
    st, tb = "model.input_blocks.1.1", "model.input_blocks.1.1.transformer_blocks.0"
    r64 = lambda t: t.to(torch.bfloat16).to(torch.float64)
    try:
        model.set_mfma_dtype("bf16")
        k_chain = None
        for k in range(4, 12):   # the first launch that goes through stchain_kernel
            n0 = eng.debug_get("n_stchain")
            eng.debug_stop_after(k)
            model(x.to(dev), ts.to(dev), c.to(dev))
            if eng.debug_get("n_stchain") > n0:
                k_chain = k
                break
        assert k_chain is not None, "the token-major bf16 schedule did not reach stchain_kernel"
        rd = lambda name: _ws(eng, name, B * seg * 192 * 2).view(torch.bfloat16).view(B, seg, 192)[:, :T].float().cpu()
        got, O = rd("tH1"), rd("tO")
        eng.debug_stop_after(k_chain - 1)
        model(x.to(dev), ts.to(dev), c.to(dev))
        xin = rd("tP")
        npart = (T + 31) // 32
        st_in = _ws(eng, "stP", B * npart * 192 * 2 * 4).view(torch.float32).view(B, npart, 192, 2).cpu()   # the block input's GroupNorm partials (of the producer's fp32 values)
    finally:
        eng.debug_stop_after(-1)
        model.set_mfma_dtype("fp32")
    W = lambda n: sd[tb + n]
    lin = lambda a, w: F.linear(r64(a), r64(w)).float()
    P = sd[st + ".proj_out.weight"].reshape(192, 192).double()
    PF = (P @ W(".ff.net.2.weight").double()).float()
    PB = (P @ W(".ff.net.2.bias").double() + sd[st + ".proj_out.bias"].double()).float()
    fold = lambda w, g: (w.double() * g.double()[None, :]).float()
    Wq = fold(W(".attn2.to_q.weight"), W(".norm2.weight"))
    bq = (W(".attn2.to_q.weight").double() @ W(".norm2.bias").double()).float()
    Wf = fold(W(".ff.net.0.proj.weight"), W(".norm3.weight"))
    bf = (W(".ff.net.0.proj.bias").double() + W(".ff.net.0.proj.weight").double() @ W(".norm3.bias").double()).float()
    lnn = lambda t: F.layer_norm(t, (192,), None, None, eps=1e-5)
    mask = ou.alignment_mask(1, T, T)[0]
    sp = lambda t: t.reshape(T, 6, 32).permute(1, 0, 2)
    worst_frac, worst_rms = 1.0, 0.0
    for i in (0, B - 1):
        ga, gb = _gn_coef_from_partials(st_in[i:i + 1], T, sd[st + ".norm.weight"], sd[st + ".norm.bias"], 1e-6)
        g = xin[i] * ga + gb
        x1 = lin(O[i], W(".attn1.to_out.0.weight")) + W(".attn1.to_out.0.bias") + g
        q = lin(lnn(x1), Wq) + bq
        kk = F.linear(c[i], W(".attn2.to_k.weight")).to(torch.bfloat16).float()
        vv = F.linear(c[i], W(".attn2.to_v.weight")).to(torch.bfloat16).float()
        sim = (torch.einsum("hid,hjd->hij", sp(q), sp(kk)) * 32 ** -0.5).masked_fill(mask[None], -torch.finfo(torch.float32).max)
        o2 = torch.einsum("hij,hjd->hid", sim.softmax(dim=-1), sp(vv)).permute(1, 0, 2).reshape(T, 192)
        x2 = lin(o2, W(".attn2.to_out.0.weight")) + W(".attn2.to_out.0.bias") + x1
        hv = lin(lnn(x2), Wf) + bf
        a, gate = hv.chunk(2, dim=-1)
        y = lin(a * F.gelu(gate), PF) + lin(x2, P.float()) + PB + xin[i]
        rng = float(got[i].abs().max())
        same = float((y.to(torch.bfloat16).float() == got[i]).float().mean())
        d = (got[i] - y).abs()
        rms = float(d.pow(2).mean().sqrt()) / rng
        # the same tail WITHOUT roundings (what the fp32 oracle would do with these inputs), for scale
        x1p = F.linear(O[i], W(".attn1.to_out.0.weight"), W(".attn1.to_out.0.bias")) + g
        print(f"stchain<bf16> sample {i}: {100 * same:.2f} % of the stored bf16 values ARE the rounded emulation; |stored - emulation| rms {rms:.2e} max {float(d.max()) / rng:.2e} of range "
              f"(half a bf16 ulp at the range is {2 ** -9 * 0.5:.1e}); x1 without roundings differs from the emulated x1 by {float((x1p - x1).abs().max()) / rng:.1e}")
        worst_frac, worst_rms = min(worst_frac, same), max(worst_rms, rms)
    assert worst_frac >= 0.97 and worst_rms <= 1.5e-3


def test_rgemm_convolutions_bf16_against_the_operand_rounded_evaluation_of_their_own_inputs(model, dev, sd_parts):
    """The same check for the token-major schedule's ResBlock convolutions (rgemm_kernel: GroupNorm + SiLU on the stored bf16 hidden state, bf16 operands, fp32
    accumulation, bias + time-embedding row + residual in fp32, bf16 store): first ResBlock, 16 x 600 frames, both convolutions on their own read-back inputs."""
    import torch.nn.functional as F
    sd = sd_parts[2]
    B, T = 16, 600
    seg = (T + 63) // 64 * 64
    x = synth.synth_latents(211, (B, T, 32))
    c = synth.synth_latents(212, (B, T, 768))
    ts = (torch.arange(B) * 47 + 3) % 1000
    eng = model._get_engine(B, T)
    rb = "model.input_blocks.1.0"
    bufs = {}
    try:
        model.set_mfma_dtype("bf16")
        for k, name in ((2, "tH0"), (3, "tM"), (4, "tP")):   # launch 1 prepares the time-embedding rows; conv_in, conv 1, conv 2 follow (scripts/probe_tm_stages.py)
            eng.debug_stop_after(k)
            model(x.to(dev), ts.to(dev), c.to(dev))
            bufs[name] = _ws(eng, name, B * seg * 192 * 2).view(torch.bfloat16).view(B, seg, 192)[:, :T].float().cpu()
            assert float(bufs[name].abs().max()) > 0, f"launch {k} did not write {name}: the schedule's launch order changed"
            npart = (T + 31) // 32
            bufs["s" + name] = _ws(eng, "s" + name, B * npart * 192 * 2 * 4).view(torch.float32).view(B, npart, 192, 2).cpu()
    finally:
        eng.debug_stop_after(-1)
        model.set_mfma_dtype("fp32")
    e = F.linear(F.silu(ou.time_embed(sd, ts)), sd[rb + ".emb_layers.1.weight"], sd[rb + ".emb_layers.1.bias"])
    for i in (0, B - 1):
        H0, M, P = (bufs[n][i].t()[None] for n in ("tH0", "tM", "tP"))   # (1, 192, T)
        a0, b0 = _gn_coef_from_partials(bufs["stH0"][i:i + 1], T, sd[rb + ".in_layers.0.weight"], sd[rb + ".in_layers.0.bias"], 1e-5)
        a1, b1 = _gn_coef_from_partials(bufs["stM"][i:i + 1], T, sd[rb + ".out_layers.0.weight"], sd[rb + ".out_layers.0.bias"], 1e-5)
        try:
            ou.ROUND_OPERANDS = "bf16"
            h = F.silu(H0 * a0[..., None] + b0[..., None])
            m = ou._conv1d(h, sd[rb + ".in_layers.2.weight"], None, padding=1) + (sd[rb + ".in_layers.2.bias"] + e[i])[None, :, None]
            h = F.silu(M * a1[..., None] + b1[..., None])
            p = ou._conv1d(h, sd[rb + ".out_layers.3.weight"], sd[rb + ".out_layers.3.bias"], padding=1) + H0
        finally:
            ou.ROUND_OPERANDS = None
        for name, want, got in (("conv 1 + embedding row", m, M), ("conv 2 + x", p, P)):
            same = float((want.to(torch.bfloat16).float() == got).float().mean())
            d = float((got - want).abs().max()) / float(got.abs().max())
            print(f"rgemm {name:24s} sample {i}: {100 * same:.2f} % of the stored bf16 values are the rounded emulation, max |diff| {d:.2e} of range")
            assert same >= 0.97 and d <= 8e-3


@pytest.mark.parametrize("case", ["cfg", "cfg_eta_mask", "nocfg_eta"])
def test_out_sched_tm_against_the_standalone_scheduler_on_its_own_hidden_state(model, dev, case):
    """out_sched_tm_kernel (out.0 GroupNorm + SiLU + out.2 conv on the token-major bf16 hidden state, guidance, DDIM update, eta noise, mask blend) had
    only ever been compared with another HIP route at 8.7e-2.  Here its INPUTS are read back (the last hidden state and its GroupNorm partials, after a
    loop call stopped before the last launch), the model output is recomputed from them on the CPU in float64 with the kernel's roundings emulated
    (GroupNorm'ed + SiLU'ed operand and out.2's weights rounded to bf16, fp32 accumulation replaced by exact sums), and pushed through said_ddim_step —
    the stand-alone scheduler kernel that IS bit-exact against the oracle (test_scheduler_step_bit_exact) and shares sched_math.h with the fused one.
    What remains is accumulation order (1e-6) and the rare operand whose fp32 value sat within an ulp of a bf16 rounding boundary (one product of 576
    off by 2^-8: ~2e-4): a wrong coefficient row, blend order, noise slice or guidance formula is orders of magnitude above both."""
    gs = 1.0 if case.startswith("nocfg") else 2.0
    B, T, N, k = (6 if gs == 1.0 else 3), 600, 50, 10
    Be = B * (2 if gs > 1 else 1)
    eta = 1.0 if "eta" in case else 0.0
    ctx = synth.synth_latents(970, (B, T, 768)).to(dev)
    lat = synth.synth_latents(971, (B, T, 32)).to(dev)
    sn = synth.synth_latents(972, (1, B, T, 32)).to(dev) if eta > 0 else None
    kw = {}
    if "mask" in case:
        mask = torch.zeros(B, T, 32, device=dev)
        mask[:, 100:300] = 1
        mask[:, :, :5] = 1
        kw = dict(init_latents=synth.synth_latents(973, (B, T, 32)).abs().clamp(0, 1).to(dev), edit_noise=synth.synth_latents(974, (B, T, 32)).to(dev), mask=mask)
    sch = model.noise_scheduler
    sch.set_timesteps(N)
    ts = sch.timesteps.numpy()
    coef = sch.coef_table(ts, eta)
    sd = synth.said_state_dict()
    try:
        model.set_mfma_dtype("bf16")
        eng = model._get_engine(Be, T)

        def run():
            return eng.denoise_loop(latents=lat, context=ctx, timesteps=ts[k:k + 1], coef=coef[k:k + 1], prediction_type="epsilon", guidance_scale=gs,
                                    guidance_rescale=0.0, latent_scale=1.0, step_noise=sn, **kw)[1]
        got = run().cpu()
        nodes = eng.graph_num_nodes()
        assert 24 <= nodes <= 48, "the large-batch bf16 schedule (out_sched_tm is its last node)"
        eng.debug_stop_after(nodes - 1)
        try:
            run()
            seg, npart = (T + 63) // 64 * 64, (T + 31) // 32
            hid = _ws(eng, "tP", Be * seg * 192 * 2).view(torch.bfloat16).view(Be, seg, 192)[:, :T].double().cpu()   # (Be, T, 192)
            st = _ws(eng, "stP", Be * npart * 192 * 2 * 4).view(torch.float32).view(Be, npart, 192, 2).double().cpu()
        finally:
            eng.debug_stop_after(-1)
        # GroupNorm(32 groups of 6 channels, eps 1e-5) from the partials (mean, M2 per 32-token tile and channel)
        cnt = torch.tensor([min(32, T - 32 * p) for p in range(npart)], dtype=torch.float64).view(1, npart, 1)
        mean_pc, m2_pc = st[..., 0], st[..., 1]
        gm = (cnt * mean_pc).view(Be, npart, 32, 6).sum(dim=(1, 3)) / (6 * T)                                   # (Be, 32)
        gm_c = gm.repeat_interleave(6, dim=1).view(Be, 1, 192)
        m2 = (m2_pc + cnt * (mean_pc - gm_c) ** 2).view(Be, npart, 32, 6).sum(dim=(1, 3))
        rstd = 1.0 / torch.sqrt(m2 / (6 * T) + 1e-5)
        a = (sd["denoiser.model.out.0.weight"].double().view(1, 192) * rstd.repeat_interleave(6, dim=1)).float().double()
        b = (sd["denoiser.model.out.0.bias"].double().view(1, 192) - gm.repeat_interleave(6, dim=1) * a).float().double()
        y = (hid * a.view(Be, 1, 192) + b.view(Be, 1, 192)).float().double()
        v = (y / (1.0 + torch.exp(-y))).float().bfloat16().double()                                              # the MFMA operand
        w = sd["denoiser.model.out.2.weight"].bfloat16().double()
        eps = torch.nn.functional.conv1d(v.transpose(1, 2), w, sd["denoiser.model.out.2.bias"].double(), padding=1).transpose(1, 2).float()   # (Be, T, 32)
        e_u, e_c = (eps[:B], eps[B:]) if gs > 1 else (None, eps)
        want = eng.ddim_step(e_c.to(dev), lat, coef[k], "epsilon", eps_uncond=None if e_u is None else e_u.to(dev), guidance_scale=gs,
                             step_noise=None if sn is None else sn[0], init_latents=kw.get("init_latents"), edit_noise=kw.get("edit_noise"), mask=kw.get("mask")).cpu()
    finally:
        model.set_mfma_dtype("fp32")
    d = (got - want).abs()
    q = float((d <= 2e-5).double().mean())
    print(f"out_sched_tm vs emulated operands + said_ddim_step [{case}]: max {float(d.max()):.2e}, {100 * q:.2f} % of elements within 2e-5, eps range {float(eps.abs().max()):.2f}")
    assert float(d.max()) <= 6e-3 and q >= 0.95   # measured: max 4.4e-4 .. 2.1e-3 (guidance triples a flipped product), 97.9 .. 99.5 % within 2e-5
    if "mask" in case:   # masked elements do not depend on the model output at all: bit-exact
        m = kw["mask"].cpu().bool()
        assert torch.equal(got[m], want[m])
