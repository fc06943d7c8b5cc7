import torch, sys, os
sys.path.insert(0, os.getcwd())
from said_amd.model.diffusion import SAID_UNet1D
from said_amd.util import synth
from oracle import unet as ou
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = SAID_UNet1D(); m.load_state_dict(synth.said_state_dict(), strict=True); m.to(dev).eval()
sd_u = synth.fill_state_dict(synth.unet_param_shapes())
B, T = 40, 1800
x = synth.synth_latents(71, (B, T, 32)); c = synth.synth_latents(72, (B, T, 768)); ts = (torch.arange(B) * 23 + 5) % 1000
out = m(x.to(dev), ts.to(dev), c.to(dev)).cpu()
for i in (0, 39):
    ref = ou.unet1d_forward(sd_u, x[i:i+1], ts[i:i+1], c[i:i+1])
    print(f"fp32 B={B} T={T} sample {i}: err {float((out[i:i+1]-ref).abs().max())/float(ref.abs().max()):.2e}")
m.set_mfma_dtype("bf16")
o16 = m(x.to(dev), ts.to(dev), c.to(dev)).cpu()
print("bf16 vs fp32 rel:", float((o16-out).abs().max())/float(out.abs().max()), "finite", bool(torch.isfinite(o16).all()))
