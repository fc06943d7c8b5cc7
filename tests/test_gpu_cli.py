"""GPU tests of the two drop-in scripts and of the feature_dim > 0 variant."""
import importlib.util
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from oracle import pipeline as op
from said_amd.util import synth

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "script", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_wav(path, n, seed):
    w = (synth.synth_waveform(seed, n).numpy() * 32767 * 3).clip(-32768, 32767).astype(np.int16)
    wavfile.write(path, 16000, w)
    return w.astype(np.float32) / 32768.0


def test_inference_cli_end_to_end(tmp_path):
    """script/inference.py on a 0.75 s WAV (not a multiple of the frame divisor): CSV layout + values vs the oracle
    (the start noise is drawn on the device, so re-seed and draw it the same way for the oracle run)."""
    cli = _load("inference")
    wav_path, out_csv = str(tmp_path / "a.wav"), str(tmp_path / "o.csv")
    wav = _write_wav(wav_path, 12100, 3)
    torch.manual_seed(123)
    cli.main(["--weights_path", "synthetic", "--audio_path", wav_path, "--output_path", out_csv, "--num_steps", "8",
              "--device", "cuda:0"])
    from said_amd.util.blendshape import DEFAULT_BLENDSHAPE_CLASSES, load_blendshape_coeffs
    lines = open(out_csv).read().splitlines()
    assert lines[0].split(",") == DEFAULT_BLENDSHAPE_CLASSES
    got = load_blendshape_coeffs(out_csv)
    wf, window_len = op.fit_audio_unet(torch.from_numpy(wav), 16000, 60, 1)
    assert got.shape == (window_len, 32) == (45, 32)
    T = int(wf.shape[0] / 16000 * 60)
    torch.manual_seed(123)
    lat = torch.randn(1, T, 32, device="cuda:0").cpu()
    ref = op.inference(synth.said_state_dict(), op.process_audio(wf), init_latents=lat, num_inference_steps=8, guidance_scale=2.0)
    assert float((got - ref.result[0, :window_len]).abs().max()) <= 1e-3


def test_batch_driver_layout_and_repeats(tmp_path):
    drv = _load("test_inference")
    adir, odir = tmp_path / "audio", tmp_path / "out"
    pid = drv.PERSON_IDS_TEST[0]
    os.makedirs(adir / pid)
    _write_wav(str(adir / pid / "sentence01.wav"), 8000, 1)
    _write_wav(str(adir / pid / "sentence03.wav"), 12000, 2)   # a longer clip: the run's ONE gather pads every clip to the longest, the writer trims
    os.makedirs(adir / "not_a_test_person")
    _write_wav(str(adir / "not_a_test_person" / "sentence01.wav"), 8000, 3)
    assert [os.path.basename(p) for _, p in drv.test_audio_paths(str(adir))] == ["sentence01.wav", "sentence03.wav"]
    drv.main(["--weights_path", "synthetic", "--audio_dir", str(adir), "--output_dir", str(odir), "--num_steps", "4",
              "--num_repeats", "3", "--batch_size", "2", "--seed", "7"])
    files = sorted(os.listdir(odir / pid))
    assert files == [f"sentence0{s}-{r}.csv" for s in (1, 3) for r in range(3)]
    from said_amd.util.blendshape import load_blendshape_coeffs
    a = load_blendshape_coeffs(str(odir / pid / "sentence01-0.csv"))
    b = load_blendshape_coeffs(str(odir / pid / "sentence01-1.csv"))
    assert a.shape == (30, 32) and not torch.equal(a, b)      # repeats differ by start noise only
    c3 = load_blendshape_coeffs(str(odir / pid / "sentence03-2.csv"))
    assert c3.shape == (45, 32) and float(c3.min()) >= 0 and float(c3.max()) <= 1
    assert float(a.min()) >= 0 and float(a.max()) <= 1


def test_feature_dim_variant_vs_oracle():
    """feature_dim > 0 (diffusion.py:108-112, 228-229, 524-526): 768 -> D projection and D-wide cross-attention."""
    from said_amd.model.diffusion import SAID_UNet1D
    from said_amd.model.wav2vec2 import AudioConfig
    D = 64
    sd = {"null_cond_emb": synth.fill_tensor("null_cond_emb", (1, 1, D))}
    sd.update(synth.fill_state_dict(synth.w2v_param_shapes(2), "audio_encoder."))
    sd.update(synth.fill_state_dict(synth.unet_param_shapes(32, 32, D), "denoiser."))
    sd["audio_proj_layer.weight"] = synth.fill_tensor("audio_proj_layer.weight", (D, 768))
    sd["audio_proj_layer.bias"] = synth.fill_tensor("audio_proj_layer.bias", (D,))
    m = SAID_UNet1D(audio_config=AudioConfig(num_hidden_layers=2), feature_dim=D)
    m.load_state_dict(sd, strict=True)
    m.to("cuda:0").eval()
    proc = op.process_audio(synth.synth_waveform(4, 16000))
    emb = m.get_audio_embedding(proc.to("cuda:0"), 60).cpu()
    a_sd, u_sd, null = op.split_state_dict(sd)
    ref_emb = torch.nn.functional.linear(op.get_audio_embedding(a_sd, proc, 60), sd["audio_proj_layer.weight"], sd["audio_proj_layer.bias"])
    assert emb.shape == (1, 60, D) and float((emb - ref_emb).abs().max()) <= 2e-3
    lat = synth.synth_latents(5, (1, 60, 32))
    out = m.inference(proc.to("cuda:0"), num_inference_steps=6, guidance_scale=2.0, init_latents=lat.to("cuda:0")).result.cpu()
    ref = op.inference(sd, proc, init_latents=lat, num_inference_steps=6, guidance_scale=2.0, audio_embedding=ref_emb)
    assert float((out - ref.result).abs().max()) <= 1e-3


def test_bench_one_rank_rccl_group():
    """`bench.py --rccl_at_one`: the sharded run's collectives (all-gather, barriers, max-reduce) on a ONE-rank "nccl" (= RCCL)
    process group — the only form in which that code path runs on a single-GPU box; one JSON line, finite result, the
    gathered tensor's checksum equal to the plain single-GPU run's."""
    import json
    import subprocess
    import sys

    def run(extra):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--num_steps", "5", "--seconds", "1",
                              "--no_cpu_baseline", "--no_roofline"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    a = run(["--rccl_at_one"])
    b = run([])
    assert "one-rank RCCL group" in a["config"]["parallelism"] and b["config"]["parallelism"] == "single GPU"
    assert a["n_gpus"] == 1 and a["config"]["gathered_checksum"] == b["config"]["gathered_checksum"]


def test_gather_uneven_pad_and_trim_on_the_one_rank_rccl_group(tmp_path):
    """said_amd/shard.py::gather_uneven's pad -> all_gather_into_tensor -> trim path on an "nccl" (= RCCL) process group.  A single-GPU box
    has only a one-rank group, where the shard is never smaller than the largest: `pad_to` (a fixed per-rank capacity, as a caller with static
    shapes would use) makes the padding and trimming run around a real RCCL collective on device memory."""
    import subprocess
    import sys
    code = r"""
import os, sys, torch
sys.path.insert(0, %r)
from said_amd import shard
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(shard.free_port()))
dist = shard.init_process_group("nccl", 0, 1, torch.device("cuda", 0))
try:
    g = torch.Generator().manual_seed(5)
    local = torch.randn(3, 7, 32, generator=g).cuda()
    out = shard.gather_uneven(dist, local, [3], 1, pad_to=5)          # padded to 5 rows for the collective, trimmed back to 3
    assert out.shape == (3, 7, 32) and torch.equal(out, local), out.shape
    out = shard.gather_uneven(dist, local, [3], 1)                    # no padding needed
    assert torch.equal(out, local)
    empty = shard.gather_uneven(dist, local[:0], [0], 1, pad_to=2)    # an empty shard
    assert empty.shape == (0, 7, 32)
    try:
        shard.gather_uneven(dist, local, [3], 1, pad_to=2)
        raise SystemExit("pad_to below the shard size must be refused")
    except ValueError:
        pass
    print("OK")
finally:
    dist.destroy_process_group()
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]
