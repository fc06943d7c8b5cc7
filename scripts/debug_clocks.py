"""Bring-up aid: per-phase shader-clock stamps of every GEMM launch of one UNet evaluation.

The stamp sites are compiled out of the product library; build it with them first:
    SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from said_amd import _engine
from said_amd.util import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda:0")
sd = {"denoiser." + k: v for k, v in synth.fill_state_dict(synth.unet_param_shapes()).items()}
sd["null_cond_emb"] = synth.fill_tensor("null_cond_emb", (1, 1, 768))
eng = _engine.Engine(dev, max(B, 2), max(T, 64)); eng.load_weights(sd)
x = synth.synth_latents(1, (B, T, 32)).to(dev); c = synth.synth_latents(2, (B, T, 768)).to(dev)
ts = torch.tensor([500] * B)
eng.unet_forward(x, ts, c)
eng.debug_clocks(True)
eng.unet_forward(x, ts, c)
clk = eng.debug_clocks(False, read=True)
DETAIL = tuple(int(v) for v in os.environ.get("CLK_DETAIL", "1,3").split(","))   # launches whose per-wave stamps are printed
labels = ["issue", "gn-fin", "ln-stat", "band-ld", "stage", "mma", "bar1", "ldsw+bar2", "epi"]
# stchain_kernel's stamps (stchain.hip clk_stamp_c; workgroup (tile 8, last sample, slice 0)).  Column owners (waves 0-5) stamp slots
#   0 entry, 12 requests issued, 14 attention tile staged, 1 operands staged + barrier, 2 to_out1 done, 3 LayerNorm2 exchanged, 4 to_q done, 5 band done,
#   6 to_out2 done, 7 LayerNorm3 exchanged, 11 / 13 / 15 GEGLU pair 0 / 1 / 2 done, 8 GEGLU barrier, 9 folded proj_out done, 10 stores done;
# helper waves (6, 7): 0 entry, 1 first barrier (window requested), 2 window parked, 3 ring primed, 4-6 the owners' three barriers, 7 GEGLU starts, 11 / 13 / 15, 8, 10.
# A slot a role never writes holds the previous launch's value or 0: only the slots of the role's own chain are differenced.
OWNER_CHAIN = [(0, "entry"), (12, "requests"), (14, "o-staged"), (1, "staged+bar"), (2, "to_out1"), (3, "LN2"), (4, "to_q"), (5, "band"), (6, "to_out2"), (7, "LN3"),
               (11, "geglu0"), (13, "geglu1"), (15, "geglu2"), (8, "geglu-bar"), (9, "ffproj"), (10, "stores")]
HELPER_CHAIN = [(0, "entry"), (1, "kv-req+bar"), (2, "kv-parked"), (3, "ring"), (4, "bar-LN2"), (5, "bar-q"), (6, "bar-band"), (7, "bar-LN3"),
                (11, "geglu0"), (13, "geglu1"), (15, "geglu2"), (8, "geglu-bar"), (10, "ffproj-half")]


def chain_line(st, waves, chain):
    base = min(int(st[w, 0]) for w in waves)
    out, prev = [], None
    for slot, name in chain:
        vals = [int(st[w, slot]) - base for w in waves if base <= int(st[w, slot]) < base + (1 << 24)]   # (stale or unwritten slots fall outside the launch's window)
        if not vals:
            continue
        t = sum(vals) / len(vals)
        out.append(f"{name}:+{int(t - prev) if prev is not None else int(t)}")
        prev = t
    return " ".join(out), int(prev or 0)


for k in range(44):
    if clk[k, 0, 0] == 0:
        print(k, "(no stamps)"); continue
    if clk[k, 0, 12] > clk[k, 0, 0] and clk[k, 0, 12] - clk[k, 0, 0] < (1 << 24):   # stchain_kernel (the only one that stamps slot 12)
        st = clk[k, :, :16]
        lo, tot_o = chain_line(st, range(6), OWNER_CHAIN)
        lh, tot_h = chain_line(st, (6, 7), HELPER_CHAIN)
        print(f"launch {k:2d} stchain_kernel, clocks since the workgroup's first stamp (mean over the role's waves; phase = time since the previous stamp)")
        print(f"      owners  (total {tot_o:6d}) {lo}")
        print(f"      helpers (total {tot_h:6d}) {lh}")
        continue
    st = clk[k, :, :10]
    base = st[:, 0].min()
    d = np.diff(st, axis=1)
    tot = st[:, 9].max() - base
    print(f"launch {k:2d} total {tot:6d} clk | " + " ".join(f"{n}:{int(d[:, i].mean()):5d}" for i, n in enumerate(labels)))
    if k in DETAIL:
        for w in range(8):
            print("      wave", w, " ".join(f"{int(v - base):6d}" for v in st[w]))
