"""Three INDEPENDENT models (no clones, no shared engine state), one host thread and one torch stream each, B clips x 600 frames, 2 steps,
four rounds: does the split-fp16 attention stay deterministic when the co-resident kernels are other instances of this engine's own step?"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from said_amd import _engine  # noqa: E402

if os.environ.get("SAID_AB_LIB"):
    _engine._LIB_PATH = os.path.abspath(os.environ["SAID_AB_LIB"])
from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.util import synth  # noqa: E402

torch.set_grad_enabled(False)
sp, B, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
T = 600
models, inputs, streams = [], [], []
for g in range(G):
    m = SAID_UNet1D()
    m.load_state_dict(synth.said_state_dict(), strict=True)
    m.to(dev).eval()
    m.clip_groups = 1
    m._get_engine(2 * B, T).debug_option("attn_split", sp)
    if os.environ.get("DET_GEMM_SPLIT") is not None:
        (m._eng if hasattr(m, "_eng") else None).debug_option("gemm_split", int(os.environ["DET_GEMM_SPLIT"]))
    models.append(m)
    inputs.append((synth.synth_latents(700 + g, (B, T, 768)).to(dev), synth.synth_latents(800 + g, (B, T, 32)).to(dev), torch.zeros(B, T * 16000 // 60, device=dev)))
    streams.append(torch.cuda.Stream(dev))
torch.cuda.synchronize()
ref = [None] * G
for rep in range(4):
    out = [None] * G

    def work(g):
        ctx, lat, wav = inputs[g]
        with torch.cuda.stream(streams[g]):
            out[g] = models[g].inference(wav, audio_embedding=ctx, num_inference_steps=2, guidance_scale=2.0, init_latents=lat).result
        streams[g].synchronize()

    th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    if rep == 0:
        ref = [o.clone() for o in out]
    print(f"attn_split={sp} {G} independent models x B={B} rep{rep}: max abs diff to rep0 per model {[float((out[g] - ref[g]).abs().max()) for g in range(G)]}", flush=True)
