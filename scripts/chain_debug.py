"""Bring-up aid for stchain.hip: guided single steps and short loops vs the CPU oracle with st_chain 0 / 1 (and ugemm_split 0)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pipeline as op, scheduler as osch, unet as ou
from said_amd.util import synth
from said_amd.model.diffusion import SAID_UNet1D
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
sd_full = synth.said_state_dict()
m = SAID_UNet1D(); m.load_state_dict(sd_full, strict=True); m.to(dev).eval()
sd_a, sd_u, null = op.split_state_dict(sd_full)

def step_err(B, Ta, opts, t=500, N=50):
    T = int(Ta / 16000 * 60)
    proc = op.process_audio([synth.synth_waveform(10 + i, Ta).numpy() for i in range(B)])
    emb = op.get_audio_embedding(sd_a, proc, T)
    lat = synth.synth_latents(100, (B, T, 32))
    o = osch.OracleDDIM(); o.set_timesteps(N)
    sch = m.noise_scheduler; sch.set_timesteps(N)
    ts = sch.timesteps.numpy(); coef = sch.coef_table(ts, 0.0)
    k = int(np.argmin(np.abs(ts - t)))
    ctx = torch.cat([null.repeat(B, T, 1), emb])
    tt = int(ts[k])
    pred = ou.unet1d_forward(sd_u, torch.cat([lat] * 2), torch.tensor([tt] * (2 * B)), ctx)
    e_u, e_c = pred.chunk(2)
    want = o.step(e_c + 2.0 * (e_c - e_u), tt, lat)
    eng = m._get_engine(2 * B, T)
    out = {}
    for name, kv in opts.items():
        for kk, vv in kv.items(): eng.debug_option(kk, vv)
        res, latf, _ = eng.denoise_loop(latents=lat.to(dev), context=emb.to(dev), timesteps=ts[k:k + 1], coef=coef[k:k + 1], prediction_type="epsilon",
                                        guidance_scale=2.0, guidance_rescale=0.0, latent_scale=1.0, step_noise=None)
        d = (latf.cpu() - want).abs()
        out[name] = (float(d.max()), [float(d[b].max()) for b in range(B)], int(eng.debug_get("n_stchain")))
    return out

def fwd_err(B, T, opts):
    x = synth.synth_latents(1, (B, T, 32)); c = synth.synth_latents(2, (B, T, 768)); ts = torch.tensor([500] * B)
    ref = ou.unet1d_forward(sd_u, x, ts, c)
    eng = m._get_engine(max(B, 2), max(T, 64))
    out = {}
    for name, kv in opts.items():
        for kk, vv in kv.items(): eng.debug_option(kk, vv)
        got = m.denoiser(x.to(dev), ts.to(dev), c.to(dev)).cpu() if hasattr(m, "denoiser") else None
        out[name] = float((got - ref).abs().max() / (ref.max() - ref.min()))
    return out

OPTS = {"chain": {"st_chain": 1, "ugemm_split": 1}, "five": {"st_chain": 0, "ugemm_split": 1}, "fp32mfma": {"st_chain": 0, "ugemm_split": 0}}
for B, Ta in ((1, 16000), (2, 16000), (1, 9867), (3, 160000)):
    print("step B", B, "Ta", Ta, step_err(B, Ta, OPTS), flush=True)
for B, T in ((2, 60), (1, 37), (2, 600)):
    try:
        print("fwd B", B, "T", T, fwd_err(B, T, OPTS), flush=True)
    except Exception as e:
        print("fwd failed", e)
