"""Same-box A/B of a said_debug_option value on the in-situ denoising loop: python scripts/option_ab.py <option> <dtype> [B=32] [N=50] [value0 value1]
One model per value (the option is set before the first inference, so clip-group clones inherit it); alternating timed runs; the
results of the two settings are compared.  Only ONE model is alive at a time: two models with three clip groups each hold more
streams than the device has hardware queues, and the second model's groups then serialise (+16 % - that artefact was measured
as a property of the option under test before this was understood)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

opt, dt = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
N = int(sys.argv[4]) if len(sys.argv) > 4 else 50
VALS = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 1)
T = 600
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
ctx = synth.synth_latents(1, (B, T, 768)).to(dev)
lat = synth.synth_latents(2, (B, T, 32)).to(dev)
wav = torch.zeros(B, T * 16000 // 60, device=dev)
def make(v):
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    m.set_mfma_dtype(dt)
    m._get_engine(2 * B, T).debug_option(opt, v)
    return m


res = {}
for rep in range(3):
    for v in VALS:
        m = make(v)
        m.inference(wav, audio_embedding=ctx, num_inference_steps=10, guidance_scale=2.0, init_latents=lat)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            r = m.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / N * 1e3)
        res[v] = r
        print(f"{opt}={v} {dt} B={B}: {best:.4f} ms per step", flush=True)
        m._eng.close()
        del m
        torch.cuda.synchronize()
print(f"max |result({VALS[1]}) - result({VALS[0]})| = {float((res[VALS[1]] - res[VALS[0]]).abs().max()):.3e}")
