"""Round 5, VERDICT r4 item 1(c): does any result depend on workspace memory nobody wrote?  Single stream, no concurrency: every workspace buffer of the
context is filled with 0x00 / 0xFF (NaN) / 0x7F (3.4e38) before a run (said_debug_ws_fill); a result that changes with the fill value is a read of
uninitialised memory.  Covers SAID.forward (said_unet_forward) and a 2-step guided loop, small-batch and large-batch schedules, fp32 (split GEMMs off / on) and bf16."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd import _engine  # noqa: E402

if os.environ.get("SAID_AB_LIB"):
    _engine._LIB_PATH = os.path.abspath(os.environ["SAID_AB_LIB"])
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = SAID_UNet1D()
m.load_state_dict(synth.said_state_dict(), strict=True)
m.to(dev).eval()
cases = [("fp32", 0, 1, 600), ("fp32", 0, 11, 600), ("fp32", 1, 11, 600), ("bf16", 0, 1, 600), ("bf16", 0, 11, 600), ("fp32", 0, 2, 37), ("bf16", 0, 16, 333), ("fp32", 1, 16, 333)]
for dtype, split, B, T in cases:
    m.set_mfma_dtype(dtype)
    m.clip_groups = 1
    Be = 2 * B
    e = m._get_engine(Be, T)
    e.debug_option("gemm_split", split)
    x = synth.synth_latents(800, (Be, T, 32)).to(dev)
    ts = torch.full((Be,), 500, dtype=torch.long)
    ctx = synth.synth_latents(700, (Be, T, 768)).to(dev)
    lat = synth.synth_latents(801, (B, T, 32)).to(dev)
    wav = torch.zeros(B, T * 16000 // 60, device=dev)
    ref = None
    for fill in (0x00, 0xFF, 0x7F, 0x00):
        e.ws_fill(fill)
        o1 = e.unet_forward(x, ts, ctx)
        e.ws_fill(fill)
        o2 = m.inference(wav, audio_embedding=ctx[:B].contiguous(), num_inference_steps=2, guidance_scale=2.0, init_latents=lat).result
        torch.cuda.synchronize()
        if ref is None:
            ref = (o1.clone(), o2.clone())
        d1, d2 = (o1 - ref[0]).abs(), (o2 - ref[1]).abs()
        print(f"{dtype} split={split} B={B} T={T} fill=0x{fill:02X}: forward nan={int(torch.isnan(o1).sum())} differs={int((o1 != ref[0]).sum())} max={float(torch.nan_to_num(d1, nan=9e9).max()):.3e} | "
              f"loop nan={int(torch.isnan(o2).sum())} differs={int((o2 != ref[1]).sum())} max={float(torch.nan_to_num(d2, nan=9e9).max()):.3e}", flush=True)
