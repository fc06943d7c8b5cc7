"""Bring-up aid: per-phase shader-clock stamps of every GEMM launch of one UNet evaluation.

The stamp sites are compiled out of the product library; build it with them first:
    SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd import _engine
from said_amd.util import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda:0")
sd = {"denoiser." + k: v for k, v in synth.fill_state_dict(synth.unet_param_shapes()).items()}
sd["null_cond_emb"] = synth.fill_tensor("null_cond_emb", (1, 1, 768))
eng = _engine.Engine(dev, max(B, 2), max(T, 64)); eng.load_weights(sd)
x = synth.synth_latents(1, (B, T, 32)).to(dev); c = synth.synth_latents(2, (B, T, 768)).to(dev)
ts = torch.tensor([500] * B)
eng.unet_forward(x, ts, c)
eng.debug_clocks(True)
eng.unet_forward(x, ts, c)
clk = eng.debug_clocks(False, read=True)
labels = ["issue", "gn-fin", "ln-stat", "band-ld", "stage", "mma", "bar1", "ldsw+bar2", "epi"]
for k in range(44):
    if clk[k, 0, 0] == 0:
        print(k, "(no stamps)"); continue
    st = clk[k, :, :16] if k == 5 else clk[k, :, :10]
    base = st[:, 0].min()
    d = np.diff(st, axis=1)
    tot = st[:, 9].max() - base
    print(f"launch {k:2d} total {tot:6d} clk | " + " ".join(f"{n}:{int(d[:, i].mean()):5d}" for i, n in enumerate(labels)))
    if k in (1, 3, 5):
        for w in range(8):
            print("      wave", w, " ".join(f"{int(v - base):6d}" for v in st[w]))
