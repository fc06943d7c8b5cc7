"""Denoising-loop time per step as a function of the number of concurrent clip groups (SAID.inference, clip_groups = G),
batch size and precision: the measurement behind SAID._pick_clip_groups.  python scripts/clip_groups_sweep.py [T] [modes] [batches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16", "fp32"]
BATCHES = [int(b) for b in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 4, 8, 12, 16, 24, 32, 33, 48, 64]
dev = torch.device("cuda:0")
m = SAID_UNet1D()
m.load_state_dict(synth.said_state_dict(), strict=True)
m.to(dev).eval()
N = 30
for mode in MODES:
    m.set_mfma_dtype(mode)
    for B in BATCHES:
        emb = torch.randn(B, T, 768, device=dev) * 0.3
        lat = torch.randn(B, T, 32, device=dev)
        wav = torch.zeros(B, int(T / 60 * 16000), device=dev)
        row = []
        for G in (1, 2, 3, 4, None):
            if G is not None and G > B:
                continue
            m.clip_groups = G
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if rep == 0:
                    m.inference(wav, num_inference_steps=N, guidance_scale=2.0, init_latents=lat, audio_embedding=emb)
                    torch.cuda.synchronize()
                e0.record()
                m.inference(wav, num_inference_steps=N, guidance_scale=2.0, init_latents=lat, audio_embedding=emb)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / N)
            row.append((G, best))
        base = row[0][1]
        print(f"{mode} B={B:3d} T={T}: " + "  ".join(f"G={g if g else 'auto=' + str(m._pick_clip_groups(B, 2 * T))}: {ms:7.3f} ms ({ms / base - 1:+.1%})" for g, ms in row), flush=True)
m.clip_groups = None
