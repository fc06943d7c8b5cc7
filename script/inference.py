"""One audio file -> one CSV of ARKit blendshape coefficients, on an MI355X.

Command-line compatible with the reference's script/inference.py (flags and defaults: script/_common.py; output:
header of 32 blendshape names, one row per frame, only the frames covered by the un-padded audio).  Optional editing
inputs (--init_sample_path / --mask_path) and per-step dumps (--save_intermediate) behave as there.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _common  # noqa: E402
from said_amd.util.blendshape import (  # noqa: E402
    DEFAULT_BLENDSHAPE_CLASSES,
    load_blendshape_coeffs,
    save_blendshape_coeffs,
    save_blendshape_coeffs_image,
)

FLAGS = ("weights_path", "audio_path", "output_path", "output_image_path", "intermediate_dir", "prediction_type", "save_image",
         "save_intermediate", "num_steps", "strength", "guidance_scale", "guidance_rescale", "eta", "fps", "divisor_unet",
         "unet_feature_dim", "device", "init_sample_path", "mask_path")


def build_parser():
    return _common.parser_with("Speech audio to blendshape coefficients with SAiD on an MI355X", FLAGS)


def _optional_csv(path, device):
    return None if path is None else load_blendshape_coeffs(path).unsqueeze(0).to(device)


def _dump_steps(intermediates, frames, directory):
    """Step k (1 = last denoising step) -> <directory>/k.csv and k.png."""
    os.makedirs(directory, exist_ok=True)
    for k, latents in enumerate(reversed(intermediates), start=1):
        table = latents[0, :frames].cpu().numpy()
        save_blendshape_coeffs_image(table, os.path.join(directory, f"{k}.png"))
        save_blendshape_coeffs(coeffs=table, classes=DEFAULT_BLENDSHAPE_CLASSES, output_path=os.path.join(directory, f"{k}.csv"))


def main(argv=None):
    args = build_parser().parse_args(argv)
    net = _common.make_model(args)
    audio, frames = _common.prepared_audio(net, args.audio_path, args.fps, args.divisor_unet)
    with torch.no_grad():
        out = net.inference(waveform_processed=audio, init_samples=_optional_csv(args.init_sample_path, args.device),
                            mask=_optional_csv(args.mask_path, args.device), num_inference_steps=args.num_steps,
                            strength=args.strength, guidance_scale=args.guidance_scale, guidance_rescale=args.guidance_rescale,
                            eta=args.eta, save_intermediate=args.save_intermediate, show_process=True)
    table = out.result[0, :frames].cpu().numpy()
    save_blendshape_coeffs(coeffs=table, classes=DEFAULT_BLENDSHAPE_CLASSES, output_path=args.output_path)
    if args.save_image:
        save_blendshape_coeffs_image(table, args.output_image_path)
    if args.save_intermediate:
        _dump_steps(out.intermediates, frames, args.intermediate_dir)


if __name__ == "__main__":
    main()
