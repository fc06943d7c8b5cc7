"""Per-phase shader-clock stamps of the token-major-activation GEMMs (xgemm_kernel) of one UNet evaluation, workgroup 8 of every launch.
Build with the stamp sites first:  SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force
    python scripts/xgemm_clocks.py [B=32] [T=600] [dtype=bf16]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda:0")
m = SAID_UNet1D()
m.load_state_dict(synth.said_state_dict(), strict=True)
m.to(dev).eval()
m.set_mfma_dtype("bf16" if dt == "bf16" else "fp32")
x = synth.synth_latents(1, (2 * B, T, 32)).to(dev)
c = synth.synth_latents(2, (2 * B, T, 768)).to(dev)
ts = torch.tensor([500] * (2 * B)).to(dev)
eng = m._get_engine(2 * B, T)
eng.debug_option("tm_acts", 1)
m(x, ts, c)
eng.debug_option("xgemm_clk", 1)
m(x, ts, c)
torch.cuda.synchronize()
clk = eng.debug_clocks(False, read=True)        # [64 launches][8][16]
eng.debug_option("xgemm_clk", 0)
names = ["prologue", "w-prime", "k-loop 0", "epilogue 0", "k-loop 1", "epilogue 1", "k-loop 2", "epilogue 2"]
for k in range(64):
    st = clk[k, :4, :15]
    if st[0, 0] == 0:
        continue
    base = st[:, 0].min()
    last = max(int(v) for v in st.flatten() if v > 0)
    cols = [i for i in range(1, 15) if st[0, i] > 0]
    parts = []
    prev = st[:, 0]
    for i in cols:
        parts.append(f"{names[i - 1] if i - 1 < len(names) else i}: {int((st[:, i] - prev).mean()):6d}")
        prev = st[:, i]
    print(f"launch {k:2d} total {last - base:7d} clk | " + "  ".join(parts))
