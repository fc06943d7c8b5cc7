"""One bf16 (or fp32) audio-encoder pass over B clips, for rocprofv3 --kernel-trace: python scripts/audio_trace.py [B=32] [bf16|fp32].
With a results db as argv[3]: print the timeline of the LAST pass instead."""
import os
import sys

if len(sys.argv) > 3:
    import sqlite3
    con = sqlite3.connect(sys.argv[3])
    rows = con.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "conv0_stats" in r[0] or "conv0_kernel" in r[0]]
    last = rows[idx[-1]:]
    t0 = last[0][1]
    for r in last:
        name = r[0].replace("void said::", "").split("(")[0][:60]
        print(f"{(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:8.1f}  grid {r[3] // r[6]}x{r[4]}x{r[5]} wg {r[6]} lds {r[7]} vgpr {r[8]}  {name}")
    print(f"pass: {(last[-1][2] - t0) / 1e6:.3f} ms, {len(last)} kernels")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
m = SAID_UNet1D()
m.load_state_dict(synth.said_state_dict(), strict=True)
m.to(dev).eval()
m.set_mfma_dtype(mode)
wav = torch.randn(B, 160000, device=dev)
for _ in range(3):
    m.get_audio_embedding(wav, 600)
torch.cuda.synchronize()
