"""Round 5, VERDICT r4 item 1(a): WHERE do the split-fp16 GEMMs (fgemm_kernel<.., SP>) deviate when other streams of this engine run beside them?

G independent models (no clones), one host thread + stream each, run said_unet_forward (Be samples x T frames, fp32 mode, gemm_split on) stopped after
launch n (said_debug_stop_after), n ascending.  For every n the whole workspace of every model is snapshotted after a run ALONE (the reference) and after
R concurrent runs; any buffer whose bytes differ is reported with the positions of the differing words (so: which buffer = which launch's output, and
whether the damage is a whole 32 x 32 MFMA tile, a row, a column or scattered words).  Events are saved to gpurun_out/race/events.npz.

usage: race_localise.py [G=3] [Be=22] [T=600] [R=6] [n_lo=1] [n_hi=auto] ; env RACE_SPLIT=0/1 (gemm_split), RACE_ATTN=0/1, RACE_FULL=reps of the full forward first
"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from said_amd import _engine  # noqa: E402

if os.environ.get("SAID_AB_LIB"):
    _engine._LIB_PATH = os.path.abspath(os.environ["SAID_AB_LIB"])
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
argv = sys.argv[1:] + [None] * 6
G = int(argv[0] or 3)
Be = int(argv[1] or 22)
T = int(argv[2] or 600)
R = int(argv[3] or 6)
n_lo = int(argv[4] or 1)
n_hi = int(argv[5] or 0)
SPLIT = int(os.environ.get("RACE_SPLIT", "1"))
ATTN = int(os.environ.get("RACE_ATTN", "0"))
FULL = int(os.environ.get("RACE_FULL", "20"))
OUT = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "race")
os.makedirs(OUT, exist_ok=True)
dev = torch.device("cuda:0")

models, engs, inputs, streams = [], [], [], []
for g in range(G):
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    e = m._get_engine(Be, T)
    e.debug_option("gemm_split", SPLIT)
    e.debug_option("attn_split", ATTN)
    models.append(m)
    engs.append(e)
    inputs.append((synth.synth_latents(800 + g, (Be, T, 32)).to(dev), torch.full((Be,), 500 + 7 * g, dtype=torch.long),
                   synth.synth_latents(700 + g, (Be, T, 768)).to(dev)))
    streams.append(torch.cuda.Stream(dev))
torch.cuda.synchronize()
bufs = engs[0].ws_buffers()
print(f"G={G} Be={Be} T={T} R={R} gemm_split={SPLIT} attn_split={ATTN}; {len(bufs)} workspace buffers, {sum(b[2] for b in bufs) / 1e6:.0f} MB per model", flush=True)


def run_one(g, reps=1):
    x, ts, ctx = inputs[g]
    with torch.cuda.stream(streams[g]):
        for _ in range(reps):
            engs[g].unet_forward(x, ts, ctx)
    streams[g].synchronize()


def run_all(reps=1):
    th = [threading.Thread(target=run_one, args=(g, reps)) for g in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()


def snapshot(g):
    out = []
    for (i, name, nb) in bufs:
        out.append(engs[g].ws_snapshot(i, nb))
    torch.cuda.synchronize()
    return out


events = []
PRINTED = [0]


def compare(tag, n, g, ref, got):
    bad = []
    for (i, name, nb), a, b in zip(bufs, ref, got):
        if torch.equal(a, b):
            continue
        wa, wb = a[: nb // 4 * 4].view(torch.int32), b[: nb // 4 * 4].view(torch.int32)
        idx = torch.nonzero(wa != wb).flatten()
        bad.append((name, int(idx.numel())))
        if len(events) < 60:
            k = idx[:4096]
            ev = dict(tag=tag, n=n, model=g, buf=name, count=int(idx.numel()), idx=k.cpu().numpy(),
                      ref=wa[k].view(torch.float32).cpu().numpy(), got=wb[k].view(torch.float32).cpu().numpy())
            events.append(ev)
            if name in ("H0", "H1", "P", "Q", "M", "X1", "X2", "X3") and PRINTED[0] < 12:   # channel-major activation [sample][192][Tp]
                PRINTED[0] += 1
                Tp = (T + 31) // 32 * 32
                ii = ev["idx"]
                bb, cc, tt = ii // (192 * Tp), (ii // Tp) % 192, ii % Tp
                rel = np.abs(ev["got"] - ev["ref"]) / np.maximum(np.abs(ev["ref"]), 1e-6)
                print(f"   [{tag} n={n} model {g} {name}] {len(ii)} words: samples {sorted(set(bb.tolist()))} channels {cc.min()}..{cc.max()} ({len(set(cc.tolist()))} distinct) "
                      f"tokens {tt.min()}..{tt.max()} ({len(set(tt.tolist()))} distinct); |diff| max {np.abs(ev['got'] - ev['ref']).max():.3e} rel median {np.median(rel):.2e} max {rel.max():.2e}")
                # per (sample, 32-token tile, 32-channel tile) counts
                from collections import Counter
                cnt = Counter(zip(bb.tolist(), (tt // 32).tolist(), (cc // 32).tolist()))
                print("      (sample, token tile, channel tile): words -> " + ", ".join(f"{k}:{v}" for k, v in sorted(cnt.items())[:24]))
                one = sorted(cnt.items())[0][0]
                sel = (bb == one[0]) & (tt // 32 == one[1]) & (cc // 32 == one[2])
                print("      first tile: tokens-in-tile " + str(sorted(set((tt[sel] % 32).tolist()))) + " channels-in-tile " + str(sorted(set((cc[sel] % 32).tolist()))))
                print("      ref/got samples: " + ", ".join(f"{a:.6f}/{b:.6f}" for a, b in list(zip(ev['ref'][sel], ev['got'][sel]))[:8]))
                if name == "X1":   # which per-(sample, channel) constant is the offset?  GroupNorm (a, b) of the residual: gn_coef[sample][192][2]
                    gi = [i for i, (_, nm, _) in enumerate(bufs) if nm == "gn_coef"][0]
                    gc = got[gi][: Be * 384 * 4].view(torch.float32).cpu().numpy().reshape(Be, 192, 2)
                    dd = ev["got"] - ev["ref"]
                    seen = set()
                    for b_, c_, d_ in zip(bb.tolist(), cc.tolist(), dd.tolist()):
                        if (b_, c_) in seen or len(seen) >= 10:
                            continue
                        seen.add((b_, c_))
                        a0, b0 = gc[b_, c_]
                        print(f"      (sample {b_}, channel {c_}): offset {d_:+.6f} | a {a0:+.6f} b {b0:+.6f} | -b {-b0:+.6f} a-b {a0 - b0:+.6f} b(c-8)-b {gc[b_, c_ - 8, 1] - b0:+.6f} b(c+8)-b {gc[b_, min(c_ + 8, 191), 1] - b0:+.6f}")
    return bad


def sweep_point(tag, n, reps):
    for g in range(G):
        engs[g].debug_stop_after(n)
    refs = []
    for g in range(G):
        run_one(g)
        refs.append(snapshot(g))
    # alone, again: is the single-stream result reproducible at all?
    alone_bad = []
    for g in range(G):
        run_one(g)
        alone_bad += compare(tag + "-alone", n, g, refs[g], snapshot(g))
    nbad, detail = 0, {}
    for rep in range(reps):
        run_all()
        for g in range(G):
            bad = compare(tag, n, g, refs[g], snapshot(g))
            if bad:
                nbad += 1
                for name, cnt in bad:
                    detail.setdefault(name, []).append(cnt)
    return alone_bad, nbad, detail


# how many launches does a forward have?  (stop_after(n) beyond the end == the whole forward: find the first n whose alone-snapshot equals n+1's)
if FULL > 0:
    alone_bad, nbad, detail = sweep_point("full", -1, FULL)
    print(f"FULL forward: alone-repro-bad={alone_bad}  concurrent deviating (model, rep) pairs {nbad} / {FULL * G}   buffers: " +
          ", ".join(f"{k} x{len(v)} (words {min(v)}..{max(v)})" for k, v in sorted(detail.items())), flush=True)

if n_hi <= 0:
    n_hi = 90
first = None
prev_sig = None
for n in range(n_lo, n_hi + 1):
    alone_bad, nbad, detail = sweep_point("sweep", n, R)
    print(f"n={n:3d}: alone-bad={alone_bad} concurrent deviating {nbad:2d} / {R * G}  " +
          ", ".join(f"{k} x{len(v)} (words {min(v)}..{max(v)})" for k, v in sorted(detail.items())), flush=True)
    if nbad and first is None:
        first = n
    if first is not None and n >= first + 12:
        break
np.savez_compressed(os.path.join(OUT, f"events_split{SPLIT}_attn{ATTN}.npz"), events=np.array(events, dtype=object), bufs=np.array(bufs, dtype=object))
print(f"first deviating launch index n = {first}; {len(events)} events saved", flush=True)
