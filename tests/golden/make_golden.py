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


def g9_loop_control_flow():
    """G9 — the reference's OWN ``SAID_UNet1D.inference`` (said/model/diffusion.py:308-472) run on CPU.

    ``said.model.diffusion`` imports ``diffusers`` (absent here), so the two names it takes from it —
    ``DDIMScheduler`` and ``rescale_noise_cfg`` — are provided by a stub module backed by ``oracle/scheduler.py``.
    What this pins to the reference's code: the loop CONTROL FLOW — guidance order (uncond first, ``e_c + s (e_c - e_u)``),
    ``t_start``, ``timesteps[-init_timestep]``, the re-noising of ``init_latents`` with the NEXT timestep and the clean
    init on the last step, ``latents / latent_scale`` and the final clamp, the order of the random draws, the 0-dim
    timestep broadcast in ``forward`` — together with the reference's own UNet and audio encoder.  What it does NOT
    pin: the scheduler arithmetic itself, which is the oracle's restatement on both sides (fixture name says so).
    ``torch.randn`` is replaced by a queue of seeded draws so the same noise can be injected into the oracle / HIP
    path: start latents (:363-367), editing noise inside ``add_noise`` (:383-385), one draw per step for eta > 0."""
    from dataclasses import dataclass
    from oracle import scheduler as osch

    class _StubDDIM:
        def __init__(self, num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2", prediction_type="epsilon"):
            assert beta_schedule == "squaredcos_cap_v2"
            self._o = osch.OracleDDIM(num_train_timesteps, prediction_type)
            self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type)
            self.init_noise_sigma = self._o.init_noise_sigma
            self.alphas_cumprod = self._o.alphas_cumprod

        @property
        def timesteps(self):
            return self._o.timesteps

        def set_timesteps(self, n, device=None):
            self._o.set_timesteps(n)

        def scale_model_input(self, sample, timestep=None):
            return sample

        def step(self, model_output, timestep, sample, eta=0.0):
            vn = torch.randn(model_output.shape) if eta > 0 else None   # diffusers draws randn_tensor(model_output.shape) here
            return types.SimpleNamespace(prev_sample=self._o.step(model_output, int(timestep), sample, eta=eta, variance_noise=vn))

        def add_noise(self, original, noise, timesteps):
            return self._o.add_noise(original, noise, timesteps)

        def get_velocity(self, sample, noise, timesteps):
            return self._o.get_velocity(sample, noise, timesteps)

    d = types.ModuleType("diffusers")
    d.DDIMScheduler = _StubDDIM
    d.SchedulerMixin = object
    d.__path__ = []
    pp = types.ModuleType("diffusers.pipelines.stable_diffusion.pipeline_stable_diffusion")
    pp.rescale_noise_cfg = osch.rescale_noise_cfg
    for name in ("diffusers", "diffusers.pipelines", "diffusers.pipelines.stable_diffusion"):
        m = sys.modules.setdefault(name, types.ModuleType(name))
        m.__path__ = []
    sys.modules["diffusers"] = d
    sys.modules["diffusers.pipelines.stable_diffusion.pipeline_stable_diffusion"] = pp
    import_reference()
    ref_diff = importlib.import_module("said.model.diffusion")

    from transformers import Wav2Vec2Config, Wav2Vec2FeatureExtractor
    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True,
                                  return_attention_mask=False)

    class _Proc:   # the slice of Wav2Vec2Processor the reference touches (diffusion.py:95, 204-206)
        feature_extractor = fe

        def __call__(self, *a, **k):
            return fe(*a, **k)

    NL = 2   # encoder layers (a 12-layer encoder adds nothing to what G9 pins; G5 pins the encoder itself)
    sd = synth.said_state_dict(num_w2v_layers=NL)
    sd_load = {k.replace("conv.weight_g", "conv.parametrizations.weight.original0").replace(
        "conv.weight_v", "conv.parametrizations.weight.original1"): v for k, v in sd.items()}
    real_randn = torch.randn
    out = {}
    cases = {
        "cfg": dict(B=2, Ta=16000, N=20, gs=2.0),
        "nocfg_inter": dict(B=1, Ta=16000, N=8, gs=1.0, save_intermediate=True),
        "edit_mask_strength": dict(B=2, Ta=16000, N=15, gs=2.0, edit=True, strength=0.6, save_intermediate=True),
        "eta_rescale": dict(B=2, Ta=8000, N=10, gs=2.5, eta=1.0, rescale=0.7),
        "sample_pred": dict(B=1, Ta=8000, N=6, gs=2.0, pred="sample"),
        "v_pred_scaled": dict(B=1, Ta=8000, N=6, gs=2.0, pred="v_prediction", latent_scale=2.0),
        "strength0": dict(B=1, Ta=8000, N=10, gs=2.0, edit=True, strength=0.0),
    }
    for name, c in cases.items():
        B, Ta, N = c["B"], c["Ta"], c["N"]
        T = int(Ta / 16000 * 60)
        m = ref_diff.SAID_UNet1D(audio_config=Wav2Vec2Config(num_hidden_layers=NL), audio_processor=_Proc(),
                                 prediction_type=c.get("pred", "epsilon"), latent_scale=c.get("latent_scale", 1))
        m.load_state_dict(sd_load, strict=True)
        m.eval()
        wav = [synth.synth_waveform(10 + i, Ta).numpy() for i in range(B)]
        proc = m.process_audio(wav)
        strength = c.get("strength", 1.0)
        init_t = min(int(N * strength), N)
        edit = c.get("edit", False)
        eta = c.get("eta", 0.0)
        # the draws the reference will make, in its order
        queue = []
        if not edit:
            queue.append(("start", synth.synth_latents(100, (B, T, 32))))
        else:
            queue.append(("edit", synth.synth_latents(102, (B, T, 32))))
        if eta > 0:
            sn = synth.synth_latents(103, (init_t, B, T, 32))
            queue += [("step", sn[k]) for k in range(init_t)]
        drawn = []

        def fake_randn(*shape, **kw):
            if len(shape) == 1 and not isinstance(shape[0], int):
                shape = tuple(shape[0])
            tag, t = queue.pop(0)
            assert tuple(t.shape) == tuple(shape), (name, tag, t.shape, shape)
            drawn.append(tag)
            return t.clone()

        kw = {}
        if edit:
            init_samples = torch.sigmoid(synth.synth_latents(101, (B, T, 32))) * 0.5
            mask = torch.zeros(B, T, 32)
            mask[:, : T // 3] = 1.0
            mask[:, :, :4] = 1.0
            kw = dict(init_samples=init_samples, mask=mask)
        torch.randn = fake_randn
        try:
            with torch.no_grad():
                o = m.inference(proc, num_inference_steps=N, strength=strength, guidance_scale=c["gs"],
                                guidance_rescale=c.get("rescale", 0.0), eta=eta, save_intermediate=c.get("save_intermediate", False), **kw)
        finally:
            torch.randn = real_randn
        assert not queue, (name, [q[0] for q in queue])
        out[name + "_result"] = o.result.numpy()
        if c.get("save_intermediate", False):
            out[name + "_inter"] = torch.stack(o.intermediates).numpy()
        print(f"  g9 {name}: draws {drawn[:3]}{'...' if len(drawn) > 3 else ''} steps {init_t} result range "
              f"[{float(o.result.min()):.3f}, {float(o.result.max()):.3f}]")
    save("g9_loop_control_flow_scheduler_leg_unpinned", **out)


def g10_vae_encoder():
    """G10 — the reference's own ``BCVAE.encode`` (said/model/vae.py:26-83, 228-243) in eval mode:
    (a) deterministic weights of said_amd.util.synth.vae_encoder_state_dict (regenerable on the GPU box);
    (b) the reference's trained weights model/vae.pth (present only in the build container: the fixture holds the
        inputs' seeds and the outputs, never the weights)."""
    import_reference()
    vae_mod = importlib.import_module("said.model.vae")
    coeffs = torch.sigmoid(synth.synth_latents(41, (5, 120, 32)))
    seq = torch.sigmoid(synth.synth_latents(42, (300, 32)))
    out = {}
    for tag, sd_enc in (("synth", synth.vae_encoder_state_dict()), ("real", None)):
        v = vae_mod.BCVAE()
        if sd_enc is None:
            v.load_state_dict(torch.load(os.path.join(REF, "model", "vae.pth"), map_location="cpu"), strict=True)
        else:
            full = v.state_dict()
            full.update(sd_enc)
            v.load_state_dict(full, strict=True)
        v.eval()
        lat = v.encode(coeffs)
        out[tag + "_mean"] = lat.mean.numpy()
        out[tag + "_log_var"] = lat.log_var.numpy()
        # the evaluation driver's sliding windows (script/test_evaluate.py:89-95), step 10 and step 1 with padding 3
        for step, pad in ((10, 0), (1, 3)):
            n = (seq.shape[0] - v.seq_len) // step + 1 - pad
            out[f"{tag}_win_s{step}_p{pad}"] = torch.stack([v.encode(seq[None, step * w: step * w + v.seq_len]).mean[0] for w in range(n)]).numpy()
    save("g10_vae_encoder", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        g9_loop_control_flow()
    elif len(sys.argv) > 1 and sys.argv[1] == "g10":
        g10_vae_encoder()
    else:
        main()
        g9_loop_control_flow()
        g10_vae_encoder()
