"""A/B of a said_debug_option on the audio encoder alone: python scripts/audio_option_ab.py <option> [B=32] [dtype=bf16] [values=0,1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

opt = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
vals = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,1").split(",")]
dev = torch.device("cuda:0")
m = SAID_UNet1D()
m.load_state_dict(synth.said_state_dict(), strict=True)
m.to(dev).eval()
m.set_mfma_dtype(dt)
wav = torch.stack([synth.synth_waveform(700 + i, 160000) for i in range(B)]).to(dev)
eng = m._get_engine(2, 64)
out = {}
for rep in range(3):
    for v in vals:
        eng.debug_option(opt, v)
        out[v] = m.get_audio_embedding(wav, 600)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m.get_audio_embedding(wav, 600)
        e1.record()
        torch.cuda.synchronize()
        print(f"{opt}={v} {dt} B={B}: {e0.elapsed_time(e1) / 5:.3f} ms per pass", flush=True)
print(f"max |out({vals[-1]}) - out({vals[0]})| = {float((out[vals[-1]] - out[vals[0]]).abs().max()):.3e}")
