"""Per-launch timing of one UNet evaluation's schedule (said_profile_unet: each stage replayed back to back, HIP events).
    python scripts/profile_stages.py [B=32] [T=600] [dtype=bf16|f32] [option=value ...]     (options: said_debug_option names)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
opts = dict(kv.split("=") for kv in sys.argv[4:])
dev = torch.device("cuda:0")
m = SAID_UNet1D()
m.load_state_dict(synth.said_state_dict(), strict=True)
m.to(dev).eval()
m.set_mfma_dtype("bf16" if dt == "bf16" else "fp32")
ctx = synth.synth_latents(1, (B, T, 768)).to(dev)
lat = synth.synth_latents(2, (B, T, 32)).to(dev)
wav = torch.zeros(B, T * 16000 // 60, device=dev)
eng = m._get_engine(2 * B, T)
for k, v in opts.items():
    eng.debug_option(k, int(v))
m.inference(wav, audio_embedding=ctx, num_inference_steps=2, guidance_scale=2.0, init_latents=lat)
torch.cuda.synchronize()
NAMES = {0: "cgemm", 1: "attn", 2: "ugemm", 4: "tgemm", 5: "prep", 6: "xgemm"}
EPI = {0: "store", 1: "qkv", 2: "geglu", 3: "band", -1: "attn", -2: "prep"}
st = eng.profile_unet(2 * B, T, reps=20, cfg_clips=B)
tot = 0.0
for i, s in enumerate(st):
    tot += s["us"]
    tf = s["flops"] / (s["us"] * 1e-6) / 1e12 if s["us"] > 0 else 0
    print(f"{i:3d} {NAMES.get(s['kind'], s['kind']):6s} {EPI.get(s['epi'], s['epi']):6s} NB={s['NB']:4d} {s['us']:9.2f} us  {s['bytes'] / 1e6:8.2f} MB  {s['bytes'] / (s['us'] * 1e-6) / 1e9:8.1f} GB/s  {tf:7.1f} TFLOP/s")
print(f"sum {tot:.1f} us over {len(st)} launches, B={B} T={T} {dt} {opts}")
