"""Inference using the SAID_UNet1D model on an MI355X.

Drop-in for the reference CLI (/root/reference/script/inference.py:17-214): same flags,
defaults and CSV output (header of 32 ARKit names, first `window_len` frames).  Differences:
the model comes from `said_amd` (hand-written HIP kernels, no CPU path), WAV decoding uses
scipy, and `--weights_path synthetic` loads the deterministic test weights (no checkpoint is
reachable offline).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.scheduler import DDIMScheduler  # noqa: E402
from said_amd.util.audio import fit_audio_unet, load_audio  # noqa: E402
from said_amd.util.blendshape import (  # noqa: E402
    DEFAULT_BLENDSHAPE_CLASSES,
    load_blendshape_coeffs,
    save_blendshape_coeffs,
    save_blendshape_coeffs_image,
)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Inference the lipsync using the SAiD model")
    p.add_argument("--weights_path", type=str, default="../BlendVOCA/SAiD.pth", help="Path of the weights of SAiD model")
    p.add_argument("--audio_path", type=str, default="../BlendVOCA/audio/FaceTalk_170731_00024_TA/sentence01.wav", help="Path of the audio file")
    p.add_argument("--output_path", type=str, default="../out.csv", help="Path of the output blendshape coefficients file (csv format)")
    p.add_argument("--output_image_path", type=str, default="../out.png", help="Path of the image of the output blendshape coefficients")
    p.add_argument("--intermediate_dir", type=str, default="../interm", help="Saving directory of the intermediate outputs")
    p.add_argument("--prediction_type", type=str, default="epsilon", help="Prediction type of the scheduler function, 'epsilon', 'sample', or 'v_prediction'")
    # NB: type=bool as in the reference (inference.py:59-70): any non-empty string parses as True.
    p.add_argument("--save_image", type=bool, default=False, help="Save the output blendshape coefficients as an image")
    p.add_argument("--save_intermediate", type=bool, default=False, help="Save the intermediate outputs")
    p.add_argument("--num_steps", type=int, default=1000, help="Number of inference steps")
    p.add_argument("--strength", type=float, default=1.0, help="How much to paint")
    p.add_argument("--guidance_scale", type=float, default=2.0, help="Guidance scale")
    p.add_argument("--guidance_rescale", type=float, default=0.0, help="Guidance scale")
    p.add_argument("--eta", type=float, default=0.0, help="Eta for DDIMScheduler, between [0, 1]")
    p.add_argument("--fps", type=int, default=60, help="FPS of the blendshape coefficients sequence")
    p.add_argument("--divisor_unet", type=int, default=1, help="Length of the blendshape coefficients sequence should be divided by this number")
    p.add_argument("--unet_feature_dim", type=int, default=-1, help="Dimension of the latent feature of the UNet")
    p.add_argument("--device", type=str, default="cuda:0", help="GPU device (MI355X); there is no CPU path")
    p.add_argument("--init_sample_path", type=str, help="Path of the initial sample file (csv format)")
    p.add_argument("--mask_path", type=str, help="Path of the mask file (csv format)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    device = args.device

    init_samples = None
    if args.init_sample_path is not None:
        init_samples = load_blendshape_coeffs(args.init_sample_path).unsqueeze(0).to(device)
    mask = None
    if args.mask_path is not None:
        mask = load_blendshape_coeffs(args.mask_path).unsqueeze(0).to(device)

    said_model = SAID_UNet1D(noise_scheduler=DDIMScheduler, feature_dim=args.unet_feature_dim, prediction_type=args.prediction_type)
    if args.weights_path == "synthetic":
        from said_amd.util import synth
        said_model.load_state_dict(synth.said_state_dict(), strict=True)
    else:
        said_model.load_state_dict(torch.load(args.weights_path, map_location="cpu"))
    said_model.to(device)
    said_model.eval()

    waveform = load_audio(args.audio_path, said_model.sampling_rate)
    fit_output = fit_audio_unet(waveform, said_model.sampling_rate, args.fps, args.divisor_unet)
    waveform, window_len = fit_output.waveform, fit_output.window_size
    waveform_processed = said_model.process_audio(waveform).to(device)

    with torch.no_grad():
        output = said_model.inference(
            waveform_processed=waveform_processed, init_samples=init_samples, mask=mask, num_inference_steps=args.num_steps,
            strength=args.strength, guidance_scale=args.guidance_scale, guidance_rescale=args.guidance_rescale, eta=args.eta,
            save_intermediate=args.save_intermediate, show_process=True)

    result = output.result[0, :window_len].cpu().numpy()
    save_blendshape_coeffs(coeffs=result, classes=DEFAULT_BLENDSHAPE_CLASSES, output_path=args.output_path)
    if args.save_image:
        save_blendshape_coeffs_image(result, args.output_image_path)
    if args.save_intermediate:
        os.makedirs(args.intermediate_dir, exist_ok=True)
        for t, interm in enumerate(reversed(output.intermediates)):
            interm_coeffs = interm[0, :window_len].cpu().numpy()
            timestep = t + 1
            save_blendshape_coeffs_image(interm_coeffs, os.path.join(args.intermediate_dir, f"{timestep}.png"))
            save_blendshape_coeffs(coeffs=interm_coeffs, classes=DEFAULT_BLENDSHAPE_CLASSES,
                                   output_path=os.path.join(args.intermediate_dir, f"{timestep}.csv"))


if __name__ == "__main__":
    main()
