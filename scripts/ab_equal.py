"""Bit-identity of two builds of the engine: runs the UNet (fp32 and bf16 mode, a few shapes) and a short guided loop on the library given as argv[1] and saves / compares
the results with those of a previous invocation:   python scripts/ab_equal.py <lib> save|cmp <file>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd import _engine  # noqa: E402

_engine._LIB_PATH = os.path.abspath(sys.argv[1])
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
out = {}
for mode in ("fp32", "bf16"):
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    m.set_mfma_dtype(mode)
    for B, T in ((2, 600), (3, 333), (1, 37), (16, 600), (2, 1800)):
        x = synth.synth_latents(11, (B, T, 32)).to(dev)
        c = synth.synth_latents(12, (B, T, 768)).to(dev)
        ts = torch.full((B,), 500, dtype=torch.long)
        out[f"{mode}_fwd_{B}x{T}"] = m.forward(x, ts.to(dev), c).cpu()
    lat = synth.synth_latents(7, (1, 600, 32)).to(dev)
    emb = synth.synth_latents(8, (1, 600, 768)).to(dev)
    out[f"{mode}_loop"] = m.inference(torch.zeros(1, 160000, device=dev), num_inference_steps=12, guidance_scale=2.0, eta=0.0, init_latents=lat, audio_embedding=emb).result.cpu()
if sys.argv[2] == "save":
    torch.save(out, sys.argv[3])
    print("saved", len(out), "tensors")
else:
    ref = torch.load(sys.argv[3])
    bad = [k for k in out if not torch.equal(out[k], ref[k])]
    for k in out:
        print(f"{k:24s} max |diff| {float((out[k] - ref[k]).abs().max()):.3e}")
    print("BIT-IDENTICAL" if not bad else f"DIFFERENT: {bad}")
