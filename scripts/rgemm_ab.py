"""Round 4: the persistent GEMMs (rgemm.hip: register-stationary weights + helper waves; said_debug_option "rgemm") against round 3's
xgemm_kernel, bf16 mode, on one box.  (The first attempt, pgemm.hip — weight slices resident in LDS — measured slower and was removed:
`git log -- said_amd/csrc/pgemm.hip`, profiles/r04a_pgemm_clocks.txt.)
    python scripts/rgemm_ab.py [check] [time] [B=32] [N=50]
check: UNet forwards (B = 16 x T = 600, ragged B = 40 x T = 333) with rgemm off / on, against each other and against the CPU oracle,
       plus one guided step at B clips; counts of launches through either kernel family.
time:  alternating timed runs of the in-situ loop (B clips x N steps), rgemm off (round 3's clip-group policy) vs on with 1, 2, 4 groups."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

args = [a for a in sys.argv[1:]]
do_check = "check" in args or not any(a in args for a in ("check", "time"))
do_time = "time" in args or not any(a in args for a in ("check", "time"))
nums = [int(a) for a in args if a.isdigit()]
B = nums[0] if len(nums) > 0 else 32
N = nums[1] if len(nums) > 1 else 50
T = 600
OPT = "rgemm"
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)


def make(pg, groups=None):
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    m.set_mfma_dtype("bf16")
    m._get_engine(2 * B, T).debug_option(OPT, pg)
    if groups is not None:
        m.clip_groups = groups
    return m


if do_check:
    from oracle import unet as ou
    unet_sd = synth.fill_state_dict(synth.unet_param_shapes())
    outs = {}
    for pg in (0, -1):
        m = make(pg, 1)
        for Bf, Tf in ((16, 600), (40, 333), (3, 64)):
            x = synth.synth_latents(901 + Tf, (Bf, Tf, 32))
            c = synth.synth_latents(902 + Tf, (Bf, Tf, 768))
            ts = (torch.arange(Bf) * 61 + 5) % 1000
            eng = m._get_engine(Bf, Tf)
            eng.debug_option(OPT, pg)
            eng.debug_option("unet_tgemm_min_tokens", 1)
            n0p, n0x = eng.debug_get("n_" + OPT), eng.debug_get("n_xgemm")
            out = m(x.to(dev), ts.to(dev), c.to(dev)).cpu()
            print(f"{OPT}={pg} B={Bf} T={Tf}: launches {OPT} {eng.debug_get('n_' + OPT) - n0p} xgemm {eng.debug_get('n_xgemm') - n0x}", flush=True)
            outs[(pg, Bf, Tf)] = out
            if pg == -1:
                d = float((out - outs[(0, Bf, Tf)]).abs().max())
                print(f"   max |rgemm - xgemm| = {d:.3e}  (range {float(out.abs().max()):.3f})", flush=True)
                for i in (0, Bf - 1):
                    ref = ou.unet1d_forward(unet_sd, x[i:i + 1], ts[i:i + 1], c[i:i + 1])
                    e = float((out[i:i + 1] - ref).abs().max()) / float(ref.abs().max())
                    e0 = float((outs[(0, Bf, Tf)][i:i + 1] - ref).abs().max()) / float(ref.abs().max())
                    print(f"   sample {i}: rgemm {e:.2e} / xgemm {e0:.2e} of range vs oracle", flush=True)
        # one guided loop of 3 steps (shared prefix, duplicate stores, constant unconditional cross-attention)
        ctx = synth.synth_latents(1, (B, T, 768)).to(dev)
        lat = synth.synth_latents(2, (B, T, 32)).to(dev)
        wav = torch.zeros(B, T * 16000 // 60, device=dev)
        m._get_engine(2 * B, T).debug_option("unet_tgemm_min_tokens", -1)
        r = m.inference(wav, audio_embedding=ctx, num_inference_steps=3, guidance_scale=2.0, init_latents=lat).result.cpu()
        outs[(pg, "loop")] = r
        print(f"rgemm={pg} guided 3-step loop: nodes/step {m._eng.graph_num_nodes()}, finite {bool(torch.isfinite(r).all())}", flush=True)
        if pg == -1:
            print(f"   max |rgemm - xgemm| after 3 guided steps = {float((r - outs[(0, 'loop')]).abs().max()):.3e}", flush=True)
        m._eng.close()
        del m
        torch.cuda.synchronize()

if do_time:
    ctx = synth.synth_latents(1, (B, T, 768)).to(dev)
    lat = synth.synth_latents(2, (B, T, 32)).to(dev)
    wav = torch.zeros(B, T * 16000 // 60, device=dev)
    variants = [("xgemm, round-3 groups", 0, None), (OPT + ", 1 group", -1, 1), (OPT + ", 2 groups", -1, 2), (OPT + ", 4 groups", -1, 4)]
    for rep in range(2):
        for name, pg, groups in variants:
            m = make(pg, groups)
            m.inference(wav, audio_embedding=ctx, num_inference_steps=10, guidance_scale=2.0, init_latents=lat)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                m.inference(wav, audio_embedding=ctx, num_inference_steps=N, guidance_scale=2.0, init_latents=lat)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / N * 1e3)
            print(f"{name:24s} bf16 B={B}: {best:.4f} ms per step", flush=True)
            m._eng.close()
            del m
            torch.cuda.synchronize()
    # per-launch isolated replays of one UNet evaluation (said_profile_unet), rgemm on
    m = make(-1, 1)
    try:
        prof = m._get_engine(2 * B, T).profile_unet(2 * B, T, reps=20, cfg_clips=B)
        print("isolated per-launch us (rgemm on):", " ".join(f"{d['us']:.1f}" for d in prof), f"sum {sum(d['us'] for d in prof):.0f}", flush=True)
    except Exception as e:  # noqa: BLE001
        print("profile_unet failed:", e)
