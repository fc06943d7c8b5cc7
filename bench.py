"""bench.py — throughput of the SAiD denoising hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: Wav2Vec2 audio encode of B synthetic
10 s clips + `--num_steps` (1000) denoising steps of the conditional UNet1D with the CLI-default
classifier-free guidance + scheduler updates -> (B, 600, 32) blendshape coefficients (BASELINE.json
configs[1]).  Inputs (processed waveform, start noise) are resident in HBM before the timed region.
With N > 1 every rank processes its own B clips (weak scaling, no data-path collective) and one
RCCL all-gather assembles the (N*B, 600, 32) result, inside the timed region.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant UNet kernel: algorithmic bytes per launch / HIP-event-timed launch duration
  cpu_baseline  the CPU oracle (a port of the reference path, oracle/) timed on this host's cores on a
                bounded sample and extrapolated to the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 measured copy
FP32_MFMA_PEAK_TFLOPS = 157.3
EPI_NAMES = {0: "store", 1: "qkv", 2: "geglu", 3: "band", -1: "attn"}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=1, help="clips per GPU")
    p.add_argument("--seconds", type=float, default=10.0, help="clip length")
    p.add_argument("--num_steps", type=int, default=1000, help="denoising steps per clip")
    p.add_argument("--guidance_scale", type=float, default=2.0)
    p.add_argument("--eta", type=float, default=0.0)
    p.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                   help="f32 (BASELINE configs[1]) or bf16 multiplies with fp32 accumulation (configs[2]: --batch 32 --num_steps 50 --dtype bf16)")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_roofline", action="store_true")
    p.add_argument("--cpu_steps", type=int, default=40, help="UNet evaluations in the CPU-baseline sample")
    return p.parse_args()


def cpu_baseline(args, T, Ta):
    """Oracle (CPU restatement of the reference path) on the host cores: one audio encode + a bounded
    number of CFG UNet steps + scheduler updates, extrapolated to num_steps."""
    from oracle import pipeline as op
    from oracle import scheduler as osch
    from oracle import unet as ou
    from said_amd.util import synth
    torch.set_grad_enabled(False)
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    sd = synth.said_state_dict()
    sd_a, sd_u, null = op.split_state_dict(sd)
    proc = op.process_audio(synth.synth_waveform(1234, Ta))
    t0 = time.perf_counter()
    emb = op.get_audio_embedding(sd_a, proc, T)
    t_audio = time.perf_counter() - t0
    do_cfg = args.guidance_scale > 1.0
    ctx = torch.cat([null.repeat(1, T, 1), emb]) if do_cfg else emb
    sch = osch.OracleDDIM()
    sch.set_timesteps(args.num_steps)
    lat = synth.synth_latents(0, (1, T, 32))
    n = max(2, args.cpu_steps)
    ou.unet1d_forward(sd_u, torch.cat([lat] * 2) if do_cfg else lat, sch.timesteps[:1].repeat(2 if do_cfg else 1), ctx)  # warm
    t0 = time.perf_counter()
    done = 0
    for t in sch.timesteps[:n]:
        if done >= 3 and time.perf_counter() - t0 > 20.0:  # bounded sample: ~20 s of CPU work
            break
        done += 1
        x = torch.cat([lat] * 2) if do_cfg else lat
        pred = ou.unet1d_forward(sd_u, x, t.repeat(x.shape[0]), ctx)
        if do_cfg:
            e_u, e_c = pred.chunk(2)
            pred = e_c + args.guidance_scale * (e_c - e_u)
        lat = sch.step(pred, int(t), lat)
    n = done
    t_step = (time.perf_counter() - t0) / n
    total = t_audio + args.num_steps * t_step
    return {"value": round(T / total, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 clip: audio encode ({t_audio:.2f} s) + {n} of {args.num_steps} CFG UNet+scheduler steps "
                      f"({t_step * 1e3:.1f} ms each) on {cores} threads, extrapolated to {args.num_steps} steps",
            "clips_per_s": round(1.0 / total, 5)}


def roofline(model, Be, T, step_ms):
    eng = model._eng
    stages = eng.profile_unet(Be, T, reps=40)
    agg = {}
    for st in stages:
        name = (f"attn_kernel<D{32 * st['NB']},KS{st['KS']}>" if st["kind"] == 1 else
                f"{'ugemm' if st['kind'] == 2 else 'cgemm'}_kernel<NB{st['NB']},KS{st['KS']},{EPI_NAMES[st['epi']]}>")
        a = agg.setdefault(name, dict(us=0.0, bytes=0.0, flops=0.0, launches=0))
        a["us"] += st["us"]; a["bytes"] += st["bytes"]; a["flops"] += st["flops"]; a["launches"] += 1
    dom = max(agg, key=lambda k: agg[k]["us"])
    d = agg[dom]
    achieved = d["bytes"] / (d["us"] * 1e-6) / 1e9
    from said_amd import _engine
    unet_bytes = _engine.unet_algorithmic_bytes(Be, T, 4)
    unet_flops = _engine.unet_algorithmic_flops(Be, T)
    sum_us = sum(a["us"] for a in agg.values())
    out = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
           "launches_per_unet": d["launches"], "avg_launch_us": round(d["us"] / d["launches"], 3),
           "alg_bytes_per_launch": round(d["bytes"] / d["launches"]),
           "kernel_tflops": round(d["flops"] / (d["us"] * 1e-6) / 1e12, 3),
           "kernel_mfma_frac": round(d["flops"] / (d["us"] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 5),
           "unet_step": {"ms_graph_replay": round(step_ms, 4), "sum_kernel_us": round(sum_us, 2), "launches": len(stages),
                         "alg_bytes": round(unet_bytes), "alg_gflop": round(unet_flops / 1e9, 3),
                         "hbm_frac": round(unet_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "mfma_fp32_frac": round(unet_flops / (step_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 5)},
           "by_kernel": {k: {"us": round(v["us"], 2), "launches": v["launches"],
                             "GBps": round(v["bytes"] / (v["us"] * 1e-6) / 1e9, 1),
                             "TFLOPs": round(v["flops"] / (v["us"] * 1e-6) / 1e12, 2)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"])}}
    tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tf):
        try:
            tr = json.load(open(tf))
            if tr.get("kernel") == dom:
                out["traffic"] = tr.get("hbm_bytes_per_launch")
        except Exception:
            pass
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: said_amd has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from said_amd.model.diffusion import SAID_UNet1D
    from said_amd.util import synth
    torch.set_grad_enabled(False)
    B, Ta = args.batch, int(round(args.seconds * 16000))
    T = int(Ta / 16000 * 60)
    model = SAID_UNet1D()
    model.load_state_dict(synth.said_state_dict(), strict=True)
    model.to(dev).eval()
    model.set_mfma_dtype("bf16" if args.dtype == "bf16" else "fp32")
    # synthetic inputs, resident in HBM before the timed region
    wav = [synth.synth_waveform(rank * B + i, Ta).numpy() for i in range(B)]
    proc = model.process_audio(wav).to(dev)
    lat0 = synth.synth_latents(rank, (B, T, 32)).to(dev)
    gathered = torch.empty(world * B, T, 32, device=dev) if world > 1 else None

    def one_pass():
        out = model.inference(proc, num_inference_steps=args.num_steps, guidance_scale=args.guidance_scale, eta=args.eta,
                              init_latents=lat0)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out.result)
        return out.result

    for _ in range(args.warmup):
        one_pass()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_pass()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(res).all()

    if rank == 0:
        frames = world * B * T * args.steps
        Be = 2 * B if args.guidance_scale > 1.0 else B
        line = {
            "metric": "blendshape frames/sec (and clips/sec) at 1000 DDPM steps, 10 s audio",
            "value": round(frames / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "clips_per_s": round(world * B * args.steps / elapsed, 4),
            "realtime_factor": round(frames / elapsed / 60.0, 2),
            "config": {"workload": f"{B} clip(s)/GPU x {args.seconds:g} s synthetic audio (T={T} frames), audio encode + "
                                   f"{args.num_steps} DDIM steps (eta={args.eta:g}), guidance_scale={args.guidance_scale:g} "
                                   f"(UNet batch {Be}), " + ("fp32; BASELINE.json configs[1]" if args.dtype == "f32" else
                                                              "bf16 multiplies / fp32 accumulation and storage; BASELINE.json configs[2] shape"),
                       "batch_per_gpu": B, "frames": T, "num_steps": args.num_steps, "guidance_scale": args.guidance_scale,
                       "eta": args.eta, "parallelism": f"clips sharded over {world} GPU(s), one RCCL all-gather" if world > 1 else "single GPU",
                       "graph_nodes_per_step": model._eng.graph_num_nodes()},
        }
        if not args.no_roofline:
            # pure denoising step time: time the loop alone (audio embedding precomputed)
            emb = model.get_audio_embedding(proc, T)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            model.inference(proc, num_inference_steps=args.num_steps, guidance_scale=args.guidance_scale, eta=args.eta,
                            init_latents=lat0, audio_embedding=emb)
            ev1.record()
            torch.cuda.synchronize()
            step_ms = ev0.elapsed_time(ev1) / args.num_steps
            line["roofline"] = roofline(model, Be, T, step_ms)
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only: other ranks would sit in the final barrier
            line["cpu_baseline"] = cpu_baseline(args, T, Ta)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
