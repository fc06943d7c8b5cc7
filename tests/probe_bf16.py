import torch, sys, os
sys.path.insert(0, os.getcwd())
from said_amd.model.diffusion import SAID_UNet1D
from said_amd.util import synth
from oracle import unet as ou
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = SAID_UNet1D(); m.load_state_dict(synth.said_state_dict(), strict=True); m.to(dev).eval()
sd_u = synth.fill_state_dict(synth.unet_param_shapes())
for (B, T) in [(2, 48), (2, 600)]:
    x = synth.synth_latents(21, (B, T, 32)); c = synth.synth_latents(121, (B, T, 768)); ts = torch.tensor([999, 17])
    ref = ou.unet1d_forward(sd_u, x, ts, c)
    m.set_mfma_dtype("fp32"); o32 = m(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    m.set_mfma_dtype("bf16"); o16 = m(x.to(dev), ts.to(dev), c.to(dev)).cpu()
    sc = float(ref.abs().max())
    print(f"B={B} T={T}: fp32 err {float((o32-ref).abs().max())/sc:.2e}  bf16 err {float((o16-ref).abs().max())/sc:.2e} (rel to max {sc:.3f}); rms rel {float((o16-ref).pow(2).mean().sqrt())/float(ref.pow(2).mean().sqrt()):.2e}")
# loop
ctx = synth.synth_latents(120, (2, 600, 768)).to(dev); lat = synth.synth_latents(121, (2, 600, 32)).to(dev)
wav = torch.zeros(2, 160000, device=dev)
m.set_mfma_dtype("fp32"); r32 = m.inference(wav, audio_embedding=ctx, num_inference_steps=50, guidance_scale=2.0, init_latents=lat).result
m.set_mfma_dtype("bf16"); r16 = m.inference(wav, audio_embedding=ctx, num_inference_steps=50, guidance_scale=2.0, init_latents=lat).result
print("loop N=50 CFG: max abs diff", float((r32-r16).abs().max()), "mean abs", float((r32-r16).abs().mean()), "nodes", m._eng.graph_num_nodes(), m._eng.get_precision())
