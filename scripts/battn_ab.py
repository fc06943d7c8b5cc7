"""battn_kernel variants (said_debug_option "battn": 0 = attn_kernel on fp32 operands, 4 / 8 = query tiles per workgroup) — per-launch isolated
replays of one UNet evaluation and the in-situ loop, bf16 mode, one box:  python scripts/battn_ab.py [B=32]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, N = 600, 50
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
ctx = synth.synth_latents(1, (B, T, 768)).to(dev)
lat = synth.synth_latents(2, (B, T, 32)).to(dev)
wav = torch.zeros(B, T * 16000 // 60, device=dev)
res = {}
for rep in range(2):
    for v in (0, 4, 8):
        m = SAID_UNet1D()
        m.load_state_dict(synth.said_state_dict(), strict=True)
        m.to(dev).eval()
        m.set_mfma_dtype("bf16")
        eng = m._get_engine(2 * B, T)
        eng.debug_option("battn", v)
        m.inference(wav, audio_embedding=ctx, num_inference_steps=10, guidance_scale=2.0, init_latents=lat)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            r = m.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / N * 1e3)
        res[v] = r
        prof = eng.profile_unet(2 * B, T, reps=20, cfg_clips=B)
        att = [d["us"] for d in prof if d["kind"] in (1, 9)]
        print(f"battn={v}: {best:.4f} ms per step; attention launches (us): " + " ".join(f"{u:.1f}" for u in att) + f"; sum of all launches {sum(d['us'] for d in prof):.0f}", flush=True)
        m._eng.close()
        del m
        torch.cuda.synchronize()
print(f"max |result(battn 8) - result(attn_kernel)| after {N} steps = {float((res[8] - res[0]).abs().max()):.3e}; 4 vs 8: {float((res[8] - res[4]).abs().max()):.3e}")
