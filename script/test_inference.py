"""Generate the inference results from test audio — batched driver on an MI355X.

Drop-in for /root/reference/script/test_inference.py:17-206: same flags and defaults, same enumeration of
`<audio_dir>/<person_id>/sentenceNN.wav` for the BlendVOCA test persons (script/dataset/dataset_voca.py:90-95,
202-215), same repeat-one-clip batching (`num_repeats` outputs per clip in chunks of `batch_size`, the batch being
the same processed waveform repeated; diversity comes from the start noise only), same `torch.manual_seed(seed)`
handling and output naming `<output_dir>/<person_id>/sentenceNN-<repeat>.csv`.  The trimesh/torchaudio-dependent
dataset classes are not needed: only the path enumeration and the 16 kHz WAV reader are on this path.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from said_amd.model.diffusion import SAID_UNet1D  # noqa: E402
from said_amd.scheduler import DDIMScheduler  # noqa: E402
from said_amd.util.audio import fit_audio_unet, load_audio  # noqa: E402
from said_amd.util.blendshape import DEFAULT_BLENDSHAPE_CLASSES, save_blendshape_coeffs  # noqa: E402

PERSON_IDS_TEST = ["FaceTalk_170731_00024_TA", "FaceTalk_170809_00138_TA"]   # dataset_voca.py:90-93
SENTENCE_IDS = list(range(1, 41))                                             # dataset_voca.py:95


def test_audio_paths(audio_dir: str, person_ids=None):
    """(person_id, audio_path) for every existing sentence file, in the reference's order."""
    out = []
    for pid in (person_ids or PERSON_IDS_TEST):
        for sid in SENTENCE_IDS:
            path = os.path.join(audio_dir, pid, f"sentence{sid:02}.wav")
            if os.path.exists(path):
                out.append((pid, path))
    return out


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Generate the inference outputs using BlendVOCA test dataset")
    p.add_argument("--weights_path", type=str, default="../BlendVOCA/SAiD.pth", help="Path of the weights of SAiD model")
    p.add_argument("--audio_dir", type=str, default="../BlendVOCA/audio", help="Directory of the audio data")
    p.add_argument("--output_dir", type=str, default="../output-inference", help="Directory of the outputs")
    p.add_argument("--prediction_type", type=str, default="epsilon", help="Prediction type of the scheduler function, 'epsilon', 'sample', or 'v_prediction'")
    p.add_argument("--num_steps", type=int, default=1000, help="Number of inference steps")
    p.add_argument("--strength", type=float, default=1.0, help="How much to paint")
    p.add_argument("--guidance_scale", type=float, default=2.0, help="Guidance scale")
    p.add_argument("--guidance_rescale", type=float, default=0.0, help="Guidance scale")
    p.add_argument("--eta", type=float, default=0.0, help="Eta for DDIMScheduler, between [0, 1]")
    p.add_argument("--fps", type=int, default=60, help="FPS of the blendshape coefficients sequence")
    p.add_argument("--divisor_unet", type=int, default=1, help="Length of the blendshape coefficients sequence should be divided by this number")
    p.add_argument("--unet_feature_dim", type=int, default=-1, help="Dimension of the latent feature of the UNet")
    p.add_argument("--device", type=str, default="cuda:0", help="GPU device (MI355X); there is no CPU path")
    p.add_argument("--num_repeats", type=int, default=72, help="Number of repetitions in inference for each audio")
    p.add_argument("--batch_size", type=int, default=64, help="Batch size for the repetition")
    p.add_argument("--seed", type=int, default=0, help="Random seed. Set the negative value if you don't want to control the randomness")
    return p


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    device = args.device
    if args.seed >= 0:
        torch.manual_seed(args.seed)

    said_model = SAID_UNet1D(noise_scheduler=DDIMScheduler, feature_dim=args.unet_feature_dim, prediction_type=args.prediction_type)
    if args.weights_path == "synthetic":
        from said_amd.util import synth
        said_model.load_state_dict(synth.said_state_dict(), strict=True)
    else:
        said_model.load_state_dict(torch.load(args.weights_path, map_location="cpu"))
    said_model.to(device)
    said_model.eval()

    with torch.no_grad():
        for pid, audio_path in test_audio_paths(args.audio_dir):
            waveform = load_audio(audio_path, said_model.sampling_rate)
            output_filename_base = os.path.splitext(os.path.basename(audio_path))[0]
            output_file_dir = os.path.join(args.output_dir, pid)
            os.makedirs(output_file_dir, exist_ok=True)

            fit_output = fit_audio_unet(waveform, said_model.sampling_rate, args.fps, args.divisor_unet)
            waveform, window_len = fit_output.waveform, fit_output.window_size
            waveform_processed = said_model.process_audio(waveform).to(device)
            waveform_processed_batch = waveform_processed.repeat(args.batch_size, 1)

            rdx = 0
            num_chunks = math.ceil(args.num_repeats / args.batch_size)
            chunk_remainder = args.num_repeats - (num_chunks - 1) * args.batch_size
            for cdx in range(num_chunks):
                chunk_size = args.batch_size if cdx < num_chunks - 1 else chunk_remainder
                output = said_model.inference(
                    waveform_processed=waveform_processed_batch[:chunk_size], num_inference_steps=args.num_steps,
                    strength=args.strength, guidance_scale=args.guidance_scale, guidance_rescale=args.guidance_rescale,
                    eta=args.eta, show_process=False)
                results = output.result[:, :window_len].cpu().numpy()
                for sdx in range(chunk_size):
                    save_blendshape_coeffs(coeffs=results[sdx], classes=DEFAULT_BLENDSHAPE_CLASSES,
                                           output_path=os.path.join(output_file_dir, f"{output_filename_base}-{rdx}.csv"))
                    rdx += 1


if __name__ == "__main__":
    main()
