"""Same-box A/B of two builds of the engine: said_amd/lib/ab_old.so vs ab_new.so (each run in its own process).
    python scripts/ab_libs.py <lib> [B=32] [N=50] [dtype ...]      prints ms per denoise step in situ (audio embedding injected)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd import _engine  # noqa: E402

_engine._LIB_PATH = os.path.abspath(sys.argv[1])
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50
T = 600
dev = torch.device("cuda:0")
m = SAID_UNet1D()
m.load_state_dict(synth.said_state_dict(), strict=True)
m.to(dev).eval()
ctx = synth.synth_latents(1, (B, T, 768)).to(dev)
lat = synth.synth_latents(2, (B, T, 32)).to(dev)
wav = torch.zeros(B, T * 16000 // 60, device=dev)
extra = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("AB_EXTRA_STREAMS", "0")))]   # other live streams of the application
for st in extra:
    with torch.cuda.stream(st):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
if os.environ.get("AB_GROUPS"):
    m.clip_groups = int(os.environ["AB_GROUPS"])
for kv in os.environ.get("AB_OPTS", "").split():      # e.g. AB_OPTS="tm_acts=1": said_debug_option before the first inference
    k, v = kv.split("=")
    m._get_engine(2 * B, T).debug_option(k, int(v))
for dt in (sys.argv[4:] or ["bf16", "fp32"]):
    m.set_mfma_dtype(dt)
    m.inference(wav, audio_embedding=ctx, num_inference_steps=10, guidance_scale=2.0, init_latents=lat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = m.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat).result
    torch.cuda.synchronize()
    probed = m._eng.debug_get("pool_probed")
    print(f"[pool candidates probed: {probed}] {os.path.basename(sys.argv[1])} B={B} {dt}: {(time.perf_counter() - t0) / 3 / N * 1e3:.4f} ms per step, checksum {float(r.double().sum()):.6f}", flush=True)
