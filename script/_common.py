"""Pieces shared by the two command-line drivers (script/inference.py, script/test_inference.py).

The flag NAMES, TYPES and DEFAULTS are the drop-in contract with the reference CLIs
(/root/reference/script/inference.py:23-117, script/test_inference.py:23-126); they are kept in one table here.
Everything behind them is this repository's own: the model is said_amd's (HIP engine, no CPU path), WAV files are read
with scipy, and `--weights_path synthetic` selects the deterministic test weights (no checkpoint is reachable offline).
"""
import argparse
import os
import sys
from typing import Iterable, Tuple

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.scheduler import DDIMScheduler  # noqa: E402
from said_amd.util.audio import fit_audio_unet, load_audio  # noqa: E402

# flag -> (type, default, description).  `bool` flags keep argparse's type=bool behaviour of the reference
# (any non-empty string, including "False", turns them on).
FLAG_TABLE = {
    "weights_path": (str, "../BlendVOCA/SAiD.pth", "checkpoint with the SAID_UNet1D state dict, or 'synthetic' for the seeded test weights"),
    "audio_path": (str, "../BlendVOCA/audio/FaceTalk_170731_00024_TA/sentence01.wav", "input speech, WAV"),
    "audio_dir": (str, "../BlendVOCA/audio", "root of <person_id>/sentenceNN.wav inputs"),
    "output_path": (str, "../out.csv", "where the (frames x 32) coefficient table is written"),
    "output_dir": (str, "../output-inference", "root of <person_id>/sentenceNN-<repeat>.csv outputs"),
    "output_image_path": (str, "../out.png", "heat-map rendering of the coefficients (with --save_image)"),
    "intermediate_dir": (str, "../interm", "one CSV + PNG per denoising step goes here (with --save_intermediate)"),
    "prediction_type": (str, "epsilon", "what the denoiser predicts: epsilon | sample | v_prediction"),
    "save_image": (bool, False, "also render the result as an image"),
    "save_intermediate": (bool, False, "also dump the latents before every step"),
    "num_steps": (int, 1000, "denoising steps"),
    "strength": (float, 1.0, "fraction of the schedule to run when editing an initial sample"),
    "guidance_scale": (float, 2.0, "classifier-free guidance weight (<= 1 disables guidance)"),
    "guidance_rescale": (float, 0.0, "rescale_noise_cfg blend factor"),
    "eta": (float, 0.0, "DDIM stochasticity in [0, 1]"),
    "fps": (int, 60, "coefficient frames per second"),
    "divisor_unet": (int, 1, "pad the audio so that the frame count is a multiple of this"),
    "unet_feature_dim": (int, -1, "width of the projected audio features (-1: use the 768 encoder features)"),
    "device": (str, "cuda:0", "the MI355X to run on"),
    "init_sample_path": (str, None, "CSV of coefficients to start from (editing)"),
    "mask_path": (str, None, "CSV mask, 1 = keep the initial sample there"),
    "num_repeats": (int, 72, "samples to draw per audio clip"),
    "batch_size": (int, 64, "samples per batch"),
    "seed": (int, 0, "torch seed; negative = leave the generator alone"),
}


def parser_with(description: str, names: Iterable[str]) -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description=description)
    for name in names:
        typ, default, text = FLAG_TABLE[name]
        ap.add_argument("--" + name, type=typ, default=default, help=text)
    return ap


def make_model(args) -> SAID_UNet1D:
    """SAID_UNet1D on args.device with the weights named by --weights_path, in eval mode."""
    net = SAID_UNet1D(noise_scheduler=DDIMScheduler, feature_dim=args.unet_feature_dim, prediction_type=args.prediction_type)
    if args.weights_path == "synthetic":
        from said_amd.util import synth
        state = synth.said_state_dict()
    else:
        state = torch.load(args.weights_path, map_location="cpu")
    net.load_state_dict(state, strict=True)
    return net.to(args.device).eval()


def prepared_audio(net: SAID_UNet1D, path: str, fps: int, divisor: int) -> Tuple[torch.Tensor, int]:
    """(1, Ta) normalised waveform on the model's device, padded for the UNet, and the number of frames to keep."""
    fitted = fit_audio_unet(load_audio(path, net.sampling_rate), net.sampling_rate, fps, divisor)
    return net.process_audio(fitted.waveform).to(next(net.parameters()).device), fitted.window_size
