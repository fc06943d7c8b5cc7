"""stchain bring-up: first SpatialTransformer's output (H1) and its GroupNorm partials, chain vs five launches, forward B=2 T=60."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd.util import synth
from said_amd.model.diffusion import SAID_UNet1D
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
m = SAID_UNet1D(); m.load_state_dict(synth.said_state_dict(), strict=True); m.to(dev).eval()
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 60
Tp = (T + 31) // 32 * 32
x = synth.synth_latents(1, (B, T, 32)).to(dev); c = synth.synth_latents(2, (B, T, 768)).to(dev); ts = torch.tensor([500] * B).to(dev)
eng = m._get_engine(max(B, 2), max(T, 64))
res = {}
for name, ch, stop in (("chain", 1, 6), ("five", 0, 10)):
    eng.debug_option("st_chain", ch)
    eng.debug_option("st_chain_dbg", 1)
    eng.debug_stop_after(stop)
    m.denoiser(x, ts, c)
    res[name] = (eng.debug_read("H1", (B, 192, Tp)).copy(), eng.debug_read("stH1", (B, Tp // 32, 192, 2)).copy(), eng.debug_read("P", (B, 192, Tp)).copy(), eng.debug_read("O", (B, 2, 192, Tp)).copy(), eng.debug_read("X1", (B, 192, Tp)).copy(), eng.debug_read("X2", (B, 192, Tp)).copy(), eng.debug_read("X3", (B, 192, Tp)).copy())
    eng.debug_stop_after(-1)
a, b = res["chain"], res["five"]
print("P (ST input) identical:", np.array_equal(a[2], b[2]), " O identical:", np.array_equal(a[3][:, 0], b[3][:, 0]))
h1a, h1b = a[0][:, :, :T], b[0][:, :, :T]
d = np.abs(h1a - h1b)
print("H1 range", h1b.min(), h1b.max(), "max abs diff", d.max(), "rel to range", d.max() / (h1b.max() - h1b.min()), "mean abs diff", d.mean())
bi, ci, ti = np.unravel_index(d.argmax(), d.shape)
print("worst at sample", bi, "channel", ci, "token", ti, h1a[bi, ci, ti], h1b[bi, ci, ti])
print("per-token max diff:", np.round(d.max(axis=(0, 1)) * 1e6, 1))
print("per-channel-tile max diff:", [float(np.round(d[:, 32 * j:32 * j + 32].max() * 1e6, 1)) for j in range(6)])
sd = np.abs(a[1] - b[1])
print("stats max diff mean:", sd[..., 0].max(), "M2:", sd[..., 1].max(), "M2 scale", np.abs(b[1][..., 1]).max())

for nm, k in (("x1", 4), ("x2", 5)):
    dd = np.abs(a[k][:, :, :T] - b[k][:, :, :T])
    print(nm, "max abs diff", dd.max(), "scale", np.abs(b[k][:, :, :T]).max(), "per-token (1e-6):", np.round(dd.max(axis=(0, 1)) * 1e6, 1))

o2a, o2b = a[6][:, :, :T], b[3][:, 0, :, :T]
dd = np.abs(o2a - o2b)
print("o2 max abs diff", dd.max(), "scale", np.abs(o2b).max(), "per-token (1e-6):", np.round(dd.max(axis=(0, 1)) * 1e6, 1))
bi, ci, ti = np.unravel_index(dd.argmax(), dd.shape)
print("worst o2 at", bi, ci, ti, o2a[bi, ci, ti], o2b[bi, ci, ti], "head", ci // 32)
print("per-head max diff at that token:", [float(np.round(dd[bi, 32 * j:32 * j + 32, ti].max() * 1e6, 1)) for j in range(6)])


x1 = a[4][:, :, :T].astype(np.float64)
mu = x1.mean(axis=1, keepdims=True); var = x1.var(axis=1, keepdims=True)
yref = (x1 - mu) / np.sqrt(var + 1e-5)

ya = a[6][:, :, :T]
dy = np.abs(ya - yref)
print("planes (read back) vs float64 LN: max", dy.max(), "per-token (1e-6):", np.round(dy.max(axis=(0, 1)) * 1e6, 1))
bi, ci, ti = np.unravel_index(dy.argmax(), dy.shape)
print("worst at sample", bi, "channel", ci, "token", ti, ya[bi, ci, ti], yref[bi, ci, ti])
bad = np.argwhere(dy > 2e-6)
print("bad elements (sample, channel, token, got, want):", [(int(b_), int(c_), int(t_), float(ya[b_, c_, t_]), float(yref[b_, c_, t_])) for b_, c_, t_ in bad[:40]])
