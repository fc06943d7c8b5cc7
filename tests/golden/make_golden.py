"""Generate the golden vectors under tests/golden/ by importing the REFERENCE.

Runs only in the build container (needs /root/reference; the GPU box never has
it).  It imports the reference's own modules —
``said.model.unet_1d_condition``, ``said.model.ldm.*``, ``said.model.wav2vec2``,
``said.util.blendshape``, ``said.util.audio`` (with ``librosa``/``torchaudio``
stubbed: they are imported at module top but unused by ``fit_audio_unet``) —
loads the deterministic name-keyed weights of ``said_amd.util.synth`` into them,
runs them on seeded inputs and stores inputs' seeds + outputs as small ``.npz``
fixtures.  ``said.model.diffusion`` is NOT importable (``diffusers`` missing), so
no scheduler golden exists: scheduler parity is unpinned (oracle/__init__.py).

Usage:  python tests/golden/make_golden.py
"""
import importlib
import io
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from said_amd.util import synth  # noqa: E402

warnings.filterwarnings("ignore")
torch.set_grad_enabled(False)


def import_reference():
    sys.path.insert(0, REF)
    for name in ("said", "said.model", "said.util"):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, name.replace(".", "/"))]
        sys.modules[name] = m
    import transformers  # noqa: F401  (must be imported before the stubs below exist)
    from transformers import Wav2Vec2Model  # noqa: F401
    mods = {}
    for n in ("said.model.unet_1d_condition", "said.model.ldm.attention", "said.model.ldm.openaimodel",
              "said.model.ldm.util", "said.model.wav2vec2", "said.util.blendshape"):
        mods[n.split(".")[-1]] = importlib.import_module(n)
    stubs = [s for s in ("librosa", "torchaudio") if s not in sys.modules]
    for stub in stubs:
        sys.modules[stub] = types.ModuleType(stub)
    mods["audio"] = importlib.import_module("said.util.audio")
    for stub in stubs:
        del sys.modules[stub]
    return mods


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path)/1024:.1f} KiB)")


def load_unet(mods):
    u = mods["unet_1d_condition"].UNet1DConditionModel(32, 32, 768)
    sd = synth.fill_state_dict(synth.unet_param_shapes())
    u.load_state_dict(sd, strict=True)
    u.eval()
    return u, sd


def main():
    mods = import_reference()
    ldm_util, ldm_attn, oai = mods["util"], mods["attention"], mods["openaimodel"]

    # G1 — timestep_embedding (ldm/util.py:66-90)
    t = torch.tensor([0, 1, 500, 999])
    save("g1_timestep_embedding", t=t.numpy(), emb=ldm_util.timestep_embedding(t, 192).numpy())

    # G2 — alignment band (ldm/attention.py:170-189), captured from the mask handed to attn2
    bands = {}
    for (T, S) in [(8, 8), (600, 600), (1800, 1800), (600, 499), (7, 10), (48, 48), (180, 180), (10, 7)]:
        blk = ldm_attn.BasicTransformerBlock(8, 1, 8, context_dim=8, checkpoint=False)
        cap = {}
        orig = blk.attn2.forward

        def rec(x, context=None, mask=None, _o=orig, _c=cap):
            _c["mask"] = mask.clone()
            return _o(x, context=context, mask=mask)

        blk.attn2.forward = rec
        blk._forward(torch.zeros(1, T, 8), torch.zeros(1, S, 8))
        m = ~cap["mask"][0]  # True = visible
        lo = m.float().argmax(dim=1)
        cnt = m.sum(dim=1)
        # every row must be one contiguous run
        for i in range(T):
            assert m[i, lo[i]:lo[i] + cnt[i]].all() and cnt[i] == m[i].sum()
        bands[f"lo_{T}_{S}"] = lo.numpy().astype(np.int32)
        bands[f"hi_{T}_{S}"] = (lo + cnt).numpy().astype(np.int32)
    save("g2_alignment_band", **bands)

    # G3 — live blocks in isolation at T=48, weights from the UNet fill
    u, sd = load_unet(mods)
    M = u.model
    x192 = synth.synth_latents(11, (2, 192, 48))
    x384 = synth.synth_latents(12, (2, 384, 48))
    emb = synth.synth_latents(13, (2, 768))
    ctx = synth.synth_latents(14, (2, 48, 768))
    save("g3_blocks",
         res192=M.input_blocks[1][0](x192, emb).numpy(),
         res384=M.output_blocks[0][0](x384, emb).numpy(),
         st=M.input_blocks[1][1](x192, ctx).numpy(),
         time_embed=M.time_embed(ldm_util.timestep_embedding(torch.tensor([3, 977]), 192)).numpy())

    # G4 — full UNet forward (unet_1d_condition.py:51-77)
    g4 = {}
    for (B, T, seed) in [(1, 48, 21), (2, 48, 22), (1, 600, 23), (2, 600, 24), (2, 37, 25)]:
        x = synth.synth_latents(seed, (B, T, 32))
        c = synth.synth_latents(seed + 100, (B, T, 768))
        ts = torch.tensor([999, 17][:B])
        g4[f"out_B{B}_T{T}"] = u(x, ts, c).numpy()
    # context length != sample length (general band; only reachable through SAID.forward)
    x = synth.synth_latents(26, (1, 40, 32)); c = synth.synth_latents(126, (1, 25, 768))
    g4["out_B1_T40_S25"] = u(x, torch.tensor([321]), c).numpy()
    save("g4_unet", **g4)
    save("weights_checksum", unet=np.float64(synth.state_dict_checksum(sd)))

    # G5 — ModifiedWav2Vec2Model on 1 s of seeded noise, num_frames=60
    from transformers import Wav2Vec2Config, Wav2Vec2FeatureExtractor
    a = mods["wav2vec2"].ModifiedWav2Vec2Model(Wav2Vec2Config())
    sda = synth.fill_state_dict(synth.w2v_param_shapes())
    sd_load = {}
    for k, v in sda.items():
        k2 = k.replace("conv.weight_g", "conv.parametrizations.weight.original0").replace(
            "conv.weight_v", "conv.parametrizations.weight.original1")
        sd_load[k2] = v
    missing = a.load_state_dict(sd_load, strict=True)
    a.eval()
    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True,
                                  return_attention_mask=False)
    wav = synth.synth_waveform(0, 16000)
    proc = fe(wav.numpy(), sampling_rate=16000, return_tensors="pt")["input_values"]
    out = a(proc, num_frames=60)
    feats = a.feature_extractor(proc)
    # second case: 2 clips of 0.5 s, 30 frames, and a no-interpolation call
    wav2 = torch.stack([synth.synth_waveform(1, 8000), synth.synth_waveform(2, 8000)])
    proc2 = fe([w.numpy() for w in wav2], sampling_rate=16000, return_tensors="pt")["input_values"]
    out2 = a(proc2, num_frames=30)
    out3 = a(proc2[:1], num_frames=None)
    save("g5_wav2vec2", last_hidden_state=out.last_hidden_state.numpy(), conv_feats=feats.numpy(),
         lhs_b2_f30=out2.last_hidden_state.numpy(), lhs_noint=out3.last_hidden_state.numpy(),
         checksum=np.float64(synth.state_dict_checksum(sda)))

    # G6 — process_audio (diffusion.py:188-207 → HF feature extractor)
    wav6 = synth.synth_waveform(5, 4000) * 3.0 + 0.25
    proc6 = fe(wav6.numpy(), sampling_rate=16000, return_tensors="pt")["input_values"]
    proc6b = fe([wav6.numpy(), synth.synth_waveform(6, 4000).numpy()], sampling_rate=16000, return_tensors="pt")["input_values"]
    save("g6_process_audio", out=proc6.numpy(), out_list=proc6b.numpy())

    # G7 — fit_audio_unet (said/util/audio.py:42-75)
    rows = []
    for (n, fps, div) in [(48000, 60, 1), (48001, 60, 1), (160000, 60, 1), (12345, 60, 1), (12345, 30, 4), (16000, 25, 1),
                          (799, 60, 1), (800, 60, 8), (100000, 24, 3)]:
        w = torch.arange(n, dtype=torch.float32)
        r = mods["audio"].fit_audio_unet(w, 16000, fps, div)
        assert torch.equal(r.waveform[:n], w) and float(r.waveform[n:].abs().sum()) == 0.0
        rows.append((n, fps, div, r.waveform.shape[0], r.window_size))
    save("g7_fit_audio", rows=np.array(rows, dtype=np.int64))

    # G8 — CSV layout (said/util/blendshape.py:36-69)
    classes = [ln.strip() for ln in open(os.path.join(REF, "data/ARKit_blendshapes.txt")) if ln.strip()] \
        if os.path.exists(os.path.join(REF, "data/ARKit_blendshapes.txt")) else None
    coeffs = synth.synth_latents(31, (5, 32)).sigmoid().numpy()
    from oracle.pipeline import BLENDSHAPE_CLASSES
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "o.csv")
        mods["blendshape"].save_blendshape_coeffs(coeffs, BLENDSHAPE_CLASSES, p)
        text = open(p).read()
        back = mods["blendshape"].load_blendshape_coeffs(p).numpy()
    save("g8_csv", coeffs=coeffs, back=back, text=np.frombuffer(text.encode(), dtype=np.uint8))
    print("arkit list in reference data == class list:", classes == BLENDSHAPE_CLASSES if classes else "n/a")


if __name__ == "__main__":
    main()
