"""Per-phase shader-clock stamps of the persistent GEMMs (rgemm_kernel) of one UNet evaluation, workgroup 8 of every launch.
Build with the stamp sites first:  SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force
    python scripts/rgemm_clocks.py [B=32] [T=600]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda:0")
m = SAID_UNet1D()
m.load_state_dict(synth.said_state_dict(), strict=True)
m.to(dev).eval()
m.set_mfma_dtype("bf16")
x = synth.synth_latents(1, (2 * B, T, 32)).to(dev)
c = synth.synth_latents(2, (2 * B, T, 768)).to(dev)
ts = torch.tensor([500] * (2 * B)).to(dev)
eng = m._get_engine(2 * B, T)
m(x, ts, c)
eng.debug_option("xgemm_clk", 1)
m(x, ts, c)
torch.cuda.synchronize()
clk = eng.debug_clocks(False, read=True)        # [64 launches][8][16]
eng.debug_option("xgemm_clk", 0)
names = ["w+issue", "prologue"] + [f"{p}{i}" for i in range(4) for p in ("mma", "epi", "park")]
for k in range(64):
    st = clk[k, :8, :15]
    if st[0, 0] == 0:
        continue
    base = st[:, 0][st[:, 0] > 0].min()
    last = max(int(v) for v in st.flatten() if v > 0)
    if st[6, 0] > 0:   # rgemm: MFMA wave 0 and helper wave 6 — slot 1 = prologue, then per period k = 1..3: [work, B2, work, B1]
        for wv, nm in ((0, "mfma0"), (5, "mfma5"), (6, "help0"), (7, "help1")):
            row = st[wv]
            cols = [i for i in range(1, 11) if row[i] > 0]
            parts, prev = [], row[0]
            for i in cols:
                parts.append(f"s{i}:{int(row[i] - prev):6d}")
                prev = row[i]
            # slot 11 / 12: inside the prologue (MFMA wave: weight fragments in registers; helper: tables written / first barrier passed), slot 14: end
            extra = " ".join(f"[s{i} at {int(row[i] - row[0]):6d}]" for i in (11, 12) if row[i] > 0)
            print(f"launch {k:2d} {nm} (+{int(row[0] - base):5d}) | " + " ".join(parts) + " " + extra + (f" | end at {int(row[14] - base):6d}" if row[14] > 0 else ""))
        continue
    st = st[:4]
    cols = [i for i in range(1, 15) if st[0, i] > 0]
    parts = []
    prev = st[:, 0]
    for i in cols:
        parts.append(f"{names[i - 1]}: {int((st[:, i] - prev).mean()):6d}")
        prev = st[:, i]
    print(f"launch {k:2d} total(stamped) {last - base:7d} clk | " + "  ".join(parts))
