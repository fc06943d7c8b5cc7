"""Two SAID models alive in one process, both running 32-clip batches as clip groups: the second model's groups must not serialise
(the clones' streams come from one pool per device).  python scripts/two_models.py [dtype=fp32]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B, T, N = 32, 600, 30
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
ctx = synth.synth_latents(1, (B, T, 768)).to(dev)
lat = synth.synth_latents(2, (B, T, 32)).to(dev)
wav = torch.zeros(B, T * 16000 // 60, device=dev)
ms = []
for i in range(2):
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    m.set_mfma_dtype(dt)
    ms.append(m)
for rep in range(2):
    for i, m in enumerate(ms):
        for g in (1, None):
            m.clip_groups = g
            m.inference(wav, audio_embedding=ctx, num_inference_steps=10, guidance_scale=2.0, init_latents=lat)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat)
            torch.cuda.synchronize()
            print(f"model {i} {dt} clip_groups={g or 'auto'}: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per step", flush=True)
