"""The G9 cases (tests/golden/make_golden.py::g9_loop_control_flow) and their inputs, shared by the CPU oracle test
and the GPU parity test."""
import torch

from oracle import pipeline as op
from said_amd.util import synth

G9_CASES = {
    "cfg": dict(B=2, Ta=16000, N=20, gs=2.0),
    "nocfg_inter": dict(B=1, Ta=16000, N=8, gs=1.0, save_intermediate=True),
    "edit_mask_strength": dict(B=2, Ta=16000, N=15, gs=2.0, edit=True, strength=0.6, save_intermediate=True),
    "eta_rescale": dict(B=2, Ta=8000, N=10, gs=2.5, eta=1.0, rescale=0.7),
    "sample_pred": dict(B=1, Ta=8000, N=6, gs=2.0, pred="sample"),
    "v_pred_scaled": dict(B=1, Ta=8000, N=6, gs=2.0, pred="v_prediction", latent_scale=2.0),
    "strength0": dict(B=1, Ta=8000, N=10, gs=2.0, edit=True, strength=0.0),
}


def g9_inputs(c):
    """Inputs of a G9 case exactly as tests/golden/make_golden.py::g9_loop_control_flow feeds them to the reference."""
    B, Ta, N = c["B"], c["Ta"], c["N"]
    T = int(Ta / 16000 * 60)
    proc = op.process_audio([synth.synth_waveform(10 + i, Ta).numpy() for i in range(B)])
    strength = c.get("strength", 1.0)
    init_t = min(int(N * strength), N)
    kw = dict(num_inference_steps=N, strength=strength, guidance_scale=c["gs"], guidance_rescale=c.get("rescale", 0.0),
              eta=c.get("eta", 0.0), save_intermediate=c.get("save_intermediate", False))
    noise = dict(init_latents=synth.synth_latents(100, (B, T, 32)))
    if c.get("edit", False):
        init_samples = torch.sigmoid(synth.synth_latents(101, (B, T, 32))) * 0.5
        mask = torch.zeros(B, T, 32)
        mask[:, : T // 3] = 1.0
        mask[:, :, :4] = 1.0
        kw.update(init_samples=init_samples, mask=mask)
        noise["edit_noise"] = synth.synth_latents(102, (B, T, 32))
    if c.get("eta", 0.0) > 0:
        noise["step_noise"] = synth.synth_latents(103, (init_t, B, T, 32))
    return proc, kw, noise, init_t
