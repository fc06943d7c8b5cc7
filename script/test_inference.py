"""Many samples per test clip — the batched driver, on an MI355X.

Command-line compatible with the reference's script/test_inference.py:17-206 (flags and defaults: script/_common.py).
For every `<audio_dir>/<person_id>/sentenceNN.wav` of the BlendVOCA test speakers (script/dataset/dataset_voca.py:90-95,
202-215) it draws `num_repeats` samples in batches of at most `batch_size`: a batch is the same normalised waveform
repeated, so samples differ only by their start noise, which comes from torch's generator after
`torch.manual_seed(seed)`.  Sample r of sentence NN lands in `<output_dir>/<person_id>/sentenceNN-<r>.csv`.
Only the path enumeration of the dataset classes is needed here, so they are not reproduced.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _common  # noqa: E402
from said_amd.util.blendshape import DEFAULT_BLENDSHAPE_CLASSES, save_blendshape_coeffs  # noqa: E402

PERSON_IDS_TEST = ["FaceTalk_170731_00024_TA", "FaceTalk_170809_00138_TA"]   # dataset_voca.py:90-93
SENTENCE_IDS = list(range(1, 41))                                             # dataset_voca.py:95

FLAGS = ("weights_path", "audio_dir", "output_dir", "prediction_type", "num_steps", "strength", "guidance_scale", "guidance_rescale",
         "eta", "fps", "divisor_unet", "unet_feature_dim", "device", "num_repeats", "batch_size", "seed")


def test_audio_paths(audio_dir: str, person_ids=None):
    """(person_id, path) of the sentence files that exist, speakers then sentence numbers ascending."""
    speakers = person_ids or PERSON_IDS_TEST
    wanted = ((pid, os.path.join(audio_dir, pid, "sentence%02d.wav" % n)) for pid in speakers for n in SENTENCE_IDS)
    return [(pid, path) for pid, path in wanted if os.path.exists(path)]


def build_parser():
    return _common.parser_with("Repeated SAiD sampling over the BlendVOCA test audio on an MI355X", FLAGS)


def batch_sizes(total: int, limit: int):
    """Sizes of consecutive batches covering `total` samples with at most `limit` each (full batches first)."""
    full, rest = divmod(total, limit)
    return [limit] * full + ([rest] if rest else [])


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    if args.seed >= 0:
        torch.manual_seed(args.seed)
    net = _common.make_model(args)
    with torch.no_grad():
        for pid, wav_path in test_audio_paths(args.audio_dir):
            stem = os.path.splitext(os.path.basename(wav_path))[0]
            target = os.path.join(args.output_dir, pid)
            os.makedirs(target, exist_ok=True)
            audio, frames = _common.prepared_audio(net, wav_path, args.fps, args.divisor_unet)
            stacked = audio.repeat(args.batch_size, 1)
            done = 0
            for n in batch_sizes(args.num_repeats, args.batch_size):
                out = net.inference(waveform_processed=stacked[:n], num_inference_steps=args.num_steps, strength=args.strength,
                                    guidance_scale=args.guidance_scale, guidance_rescale=args.guidance_rescale, eta=args.eta,
                                    show_process=False)
                tables = out.result[:, :frames].cpu().numpy()
                for k in range(n):
                    save_blendshape_coeffs(coeffs=tables[k], classes=DEFAULT_BLENDSHAPE_CLASSES,
                                           output_path=os.path.join(target, f"{stem}-{done + k}.csv"))
                done += n


if __name__ == "__main__":
    main()
