"""Experiment: 32 clips as ONE batch on one stream vs two half-batches on two streams (two engine contexts), in situ.
    python scripts/two_streams.py [dtype=bf16] [N=50]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
B, T = 32, 600


def mk():
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    m.set_mfma_dtype("bf16" if dt == "bf16" else "fp32")
    return m


ctx = synth.synth_latents(1, (B, T, 768)).to(dev)
lat = synth.synth_latents(2, (B, T, 32)).to(dev)
wav = torch.zeros(B, T * 16000 // 60, device=dev)
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ms = [mk() for _ in range(G)]
ss = [torch.cuda.Stream() for _ in range(G)]
m0 = ms[0]


def one():
    return m0.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result


def two(parts=2):
    h = B // G
    outs = [None] * G
    for i, (m, st) in enumerate(zip(ms, ss)):
        with torch.cuda.stream(st):
            outs[i] = m.inference(wav[i * h:(i + 1) * h], audio_embedding=ctx[i * h:(i + 1) * h], num_inference_steps=N, guidance_scale=2.0,
                                  init_latents=lat[i * h:(i + 1) * h]).result
    torch.cuda.synchronize()
    return torch.cat(outs)


for name, fn in (("one batch of 32, one stream", one), (f"{G} groups, {G} streams", two), ("one batch of 32, one stream", one), (f"{G} groups, {G} streams", two)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    print(f"{dt} {name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per 32 clips x {N} steps, checksum {float(r.double().sum()):.4f}", flush=True)
