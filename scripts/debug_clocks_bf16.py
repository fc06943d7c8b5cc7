"""Bring-up aid: shader-clock stamps of stchain_kernel<bf16> inside the token-major bf16 schedule (workgroup (8, last sample): all eight waves, 16 slots).
    SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force      first (stamp sites are compiled out of the product library)
slots: 0 entry | 12 requests issued | 14 tiles parked | 1 operands staged (barrier) | 2 to_out1 | 3 LayerNorm2 + planes (barrier) | 4 to_q | 5 band + planes (barrier) | 6 to_out2 + x2 |
       7 LayerNorm3 + planes (barrier) | 11 / 13 / 15 GEGLU pairs | 8 GEGLU done (barrier) | 9 proj_out | 10 end;  helper waves 6, 7: 1 barrier, 2 window parked, 3 ring primed, 4-6 barriers"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd import _engine
from said_amd.util import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 600
dev = torch.device("cuda:0")
sd = {"denoiser." + k: v for k, v in synth.fill_state_dict(synth.unet_param_shapes()).items()}
sd["null_cond_emb"] = synth.fill_tensor("null_cond_emb", (1, 1, 768))
eng = _engine.Engine(dev, B, 640); eng.load_weights(sd); eng.set_precision(True)
x = synth.synth_latents(1, (B, T, 32)).to(dev); c = synth.synth_latents(2, (B, T, 768)).to(dev)
ts = torch.tensor([500] * B)
eng.unet_forward(x, ts, c)
eng.debug_option("xgemm_clk", 1)
eng.unet_forward(x, ts, c)
eng.debug_option("xgemm_clk", 0)
clk = eng.debug_clocks(False, read=True)
order = [0, 12, 14, 1, 2, 3, 4, 5, 6, 7, 11, 13, 15, 8, 9, 10]
for k in range(64):
    st = clk[k]
    if st[0, 0] == 0 or st[0, 10] == 0: continue
    base = st[:, 0].min()
    print(f"launch {k}: total {int(st[:, 10].max() - base)} clocks")
    print("   slot   " + " ".join(f"{o:6d}" for o in order))
    for w in range(8):
        print(f"   wave {w} " + " ".join(f"{int(st[w, o] - base) if st[w, o] else 0:6d}" for o in order))
