"""Bring-up probe: bf16 mode's first launches against the bf16-emulating oracle's trace, stage by stage (where does the emulation stop matching?)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ou
from said_amd import _engine
from said_amd.util import synth
torch.set_grad_enabled(False)
B, T = 2, 600
dev = torch.device("cuda:0")
sd_u = synth.fill_state_dict(synth.unet_param_shapes())
sd = {"denoiser." + k: v for k, v in sd_u.items()}
sd["null_cond_emb"] = synth.fill_tensor("null_cond_emb", (1, 1, 768))
eng = _engine.Engine(dev, 2, 640)
eng.load_weights(sd)
eng.set_precision("bf16")
x = synth.synth_latents(21, (B, T, 32)); c = synth.synth_latents(121, (B, T, 768)); ts = torch.tensor([999, 17])
def trace(mode):
    ou.TRACE = []; ou.ROUND_OPERANDS = mode
    try:
        ou.unet1d_forward(sd_u, x, ts, c)
        return {n: t for n, t in ou.TRACE}
    finally:
        ou.TRACE = None; ou.ROUND_OPERANDS = None
t32, t16 = trace(None), trace("bf16")
Tp = (T + 31) // 32 * 32
names = eng.stage_names() if hasattr(eng, "stage_names") else None
b0 = "model.input_blocks.1.1.transformer_blocks.0"
stages = [(1, "conv_in", "H0", 192, 192), (2, "model.input_blocks.1.0:mid", "M", 192, 192), (3, "model.input_blocks.1.0:out", "P", 192, 192),
          (5, b0 + ".attn1:attn", "O", 192, 384), (6, b0 + ":x1", "X1", 192, 192), (7, b0 + ".attn2:attn", "O", 192, 384), (8, b0 + ":x2", "X2", 192, 192),
          (9, b0 + ".ff:geglu", "F", 768, 768), (10, "model.input_blocks.1.1:out", "H1", 192, 192)]
for k, tn, buf, C, rows in stages:
    eng.debug_stop_after(k)
    eng.unet_forward(x.to(dev), ts, c.to(dev))
    a = eng.debug_read(buf, (B, rows, Tp))[:, :C, :T]
    for lab, tr in (("fp32 oracle", t32), ("emulation", t16)):
        r = tr[tn].numpy()
        if r.shape != a.shape: r = r.transpose(0, 2, 1)
        print(f"launch {k} {tn:34s} vs {lab:12s}: max {np.abs(a - r).max():.3e} rms {np.sqrt(((a - r) ** 2).mean()):.3e} (range {np.abs(r).max():.2f})")
eng.debug_stop_after(-1)
