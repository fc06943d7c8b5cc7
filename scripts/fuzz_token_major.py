"""GPU fuzz (not a test): the token-major GEMM path forced ON for every launch against the channel-major kernels forced on for
every launch, random (B, T) incl. tiny and ragged shapes, both precision modes.  Run on the GPU box:
    python scripts/fuzz_token_major.py [n_cases] [seed] [BxT,BxT,...]     (explicit shapes replace the first cases)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)


def make(min_tokens):
    os.environ["SAID_UNET_TGEMM_MIN"] = str(min_tokens)
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(torch.device("cuda:0")).eval()
    return m


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    shapes = [tuple(int(v) for v in sh.split("x")) for sh in sys.argv[3].split(",")] if len(sys.argv) > 3 else []
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    worst = {"fp32": 0.0, "bf16": 0.0}
    for case in range(n):
        B = int(torch.randint(1, 72, (1,), generator=g))
        T = int(torch.randint(5, 700, (1,), generator=g))
        edge = (5, 30, 31, 32, 33, 62, 63, 64, 65, 94, 127, 129)   # tile / padding-row boundaries first
        if case < len(edge):
            T = edge[case]
        if case < len(shapes):
            B, T = shapes[case]
        if B * T > 45000 and case >= len(shapes):
            B = max(1, 45000 // T)
        x = synth.synth_latents(1000 + case, (B, T, 32)).to(dev)
        c = synth.synth_latents(2000 + case, (B, T, 768)).to(dev)
        ts = ((torch.arange(B) * 37 + case) % 1000).to(dev)
        # engines are created lazily at first use with the environment of that moment: build and run each model in turn
        out = {}
        for name, min_tokens in (("tm", 0), ("cm", 10 ** 12)):
            m = make(min_tokens)
            for mode in ("fp32", "bf16"):
                m.set_mfma_dtype(mode)
                out[(name, mode)] = m(x, ts, c).float().cpu()
            m.set_mfma_dtype("fp32")
            del m
        for mode, tol in (("fp32", 2e-5), ("bf16", 3e-2)):
            a, b = out[("tm", mode)], out[("cm", mode)]
            scale = float(b.abs().max())
            e = float((a - b).abs().max()) / scale
            ok = bool(torch.isfinite(a).all()) and e <= tol
            worst[mode] = max(worst[mode], e)
            print(f"case {case:2d} B={B:3d} T={T:3d} {mode}: token-major vs channel-major {e:.2e} of range {'ok' if ok else 'FAIL'}", flush=True)
            if not ok:
                sys.exit(1)
    print("worst:", worst)


if __name__ == "__main__":
    main()
