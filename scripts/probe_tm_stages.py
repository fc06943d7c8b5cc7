"""Bring-up probe: which token-major buffer each launch of the bf16 large-batch schedule writes (max |value| after stopping behind launch k)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd import _engine
from said_amd.util import synth
torch.set_grad_enabled(False)
B, T = 16, 600
dev = torch.device("cuda:0")
sd = {"denoiser." + k: v for k, v in synth.fill_state_dict(synth.unet_param_shapes()).items()}
sd["null_cond_emb"] = synth.fill_tensor("null_cond_emb", (1, 1, 768))
eng = _engine.Engine(dev, B, 640); eng.load_weights(sd); eng.set_precision(True)
x = synth.synth_latents(211, (B, T, 32)).to(dev); c = synth.synth_latents(212, (B, T, 768)).to(dev); ts = (torch.arange(B) * 47 + 3) % 1000
seg = (T + 63) // 64 * 64
names = [(i, nm, nb) for i, nm, nb in eng.ws_buffers() if nm.startswith("t")]
prev = {}
for k in range(1, 9):
    eng.debug_stop_after(k)
    eng.unet_forward(x, ts, c)
    out = []
    for i, nm, nb in names:
        t = eng.ws_snapshot(i, min(nb, B * seg * 192 * 2)); torch.cuda.synchronize()
        v = t.view(torch.bfloat16).float()
        sig = (float(v.abs().max()), float(v.double().sum()))
        if prev.get(nm) != sig: out.append(f"{nm} max {sig[0]:.3f}")
        prev[nm] = sig
    print(f"launch {k}: changed -> " + ", ".join(out), "| n_stchain", eng.debug_get("n_stchain"))
eng.debug_stop_after(-1)
