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
    ap = _common.parser_with("Repeated SAiD sampling over the BlendVOCA test audio on an MI355X", FLAGS)
    # not a reference flag: the reference's batched caller is single-device (test_inference.py:103-107); see said_amd/shard.py
    ap.add_argument("--gpus", type=int, default=0,
                    help="0 (default): one device (--device), start noise from torch's generator as in the reference; N >= 1: the repeats of "
                         "every clip are partitioned over N MI355X of this node (one process per GPU, ONE RCCL all-gather at the end of the run), each "
                         "repeat's start noise seeded by (seed, sentence, repeat) so that the output does not depend on N")
    return ap


def batch_sizes(total: int, limit: int):
    """Sizes of consecutive batches covering `total` samples with at most `limit` each (full batches first)."""
    full, rest = divmod(total, limit)
    return [limit] * full + ([rest] if rest else [])


def repeat_latents(seed: int, sentence: int, repeat: int, frames: int, channels: int = 32) -> torch.Tensor:
    """Start noise of one repeat of one sentence, a pure function of (seed, sentence, repeat): the same sample whichever rank draws it."""
    g = torch.Generator().manual_seed((max(seed, 0) * 1000003 + sentence) * 1000003 + repeat)
    return torch.randn(1, frames, channels, generator=g)


def sample_repeats(net, audio, n_frames_model, repeats, args, sentence: int, seeded: bool) -> torch.Tensor:
    """(len(repeats), T, 32) results for the given repeat ids of one clip, in batches of at most --batch_size.  The clip is encoded ONCE:
    every row of a batch is the same waveform (the reference encodes the repeated batch, test_inference.py:167-168)."""
    dev = audio.device
    emb1 = net.get_audio_embedding(audio, n_frames_model)
    outs, ids = [], list(repeats)
    for n in batch_sizes(len(ids), args.batch_size):
        chunk, ids = ids[:n], ids[n:]
        kw = {}
        if seeded:
            kw["init_latents"] = torch.cat([repeat_latents(args.seed, sentence, r, n_frames_model) for r in chunk]).to(dev)
        out = net.inference(waveform_processed=audio.expand(n, -1), num_inference_steps=args.num_steps, strength=args.strength,
                            guidance_scale=args.guidance_scale, guidance_rescale=args.guidance_rescale, eta=args.eta,
                            show_process=False, audio_embedding=emb1.expand(n, -1, -1).contiguous(), **kw)
        outs.append(out.result)
    return torch.cat(outs) if outs else torch.zeros(0, n_frames_model, 32, device=dev)


def write_clip(args, pid: str, stem: str, frames: int, table) -> None:
    """One CSV per repeat of a clip: <output_dir>/<person>/<sentence>-<k>.csv, the first `frames` frames (test_inference.py:188-200)."""
    target = os.path.join(args.output_dir, pid)
    os.makedirs(target, exist_ok=True)
    for k in range(table.shape[0]):
        save_blendshape_coeffs(coeffs=table[k, :frames], classes=DEFAULT_BLENDSHAPE_CLASSES, output_path=os.path.join(target, f"{stem}-{k}.csv"))


def run_rank(args, rank: int, world: int, dist) -> None:
    """One rank (world == 1 and dist None: the plain single-device driver)."""
    from said_amd import shard
    seeded = args.gpus >= 1
    if world > 1:
        args.device = f"cuda:{rank}"
    if args.seed >= 0:
        torch.manual_seed(args.seed)
    net = _common.make_model(args)
    # The repeats of every clip are sharded over the ranks (contiguous, uneven allowed); a rank runs its repeats of ALL clips with no
    # collective in between, and ONE all-gather at the end of the run returns everything ("a single RCCL all-gather ... at the
    # end"; round 4 gathered once per clip: 80 collectives for the BlendVOCA test split).  Clips differ in length: rows are padded to the
    # longest clip for the gather and trimmed when written.
    shards = shard.shard_bounds(args.num_repeats, world)
    mine = shards[rank]
    clips, local = [], []
    with torch.no_grad():
        for si, (pid, wav_path) in enumerate(test_audio_paths(args.audio_dir)):
            stem = os.path.splitext(os.path.basename(wav_path))[0]
            audio, frames = _common.prepared_audio(net, wav_path, args.fps, args.divisor_unet)
            n_model = int(audio.shape[1] / net.sampling_rate * 60)   # frames the model generates: SAID.inference's own fps (60), as in the reference
            out = sample_repeats(net, audio, n_model, mine, args, si, seeded)
            if out.shape[0] != len(mine):
                raise RuntimeError(f"path returned {out.shape[0]} items for a shard of {len(mine)}")
            if world == 1 and dist is None:
                # single device: written as it is produced, like the reference's loop (test_inference.py:188-200) — a failure on a later clip keeps what
                # exists, and nothing but the current clip is held
                write_clip(args, pid, stem, frames, out.cpu().numpy())
                continue
            clips.append((pid, stem, frames))
            local.append(out)
        if not clips:
            return
        t_max = max(o.shape[1] for o in local)
        stack = torch.zeros(len(mine), len(clips), t_max, 32, device=local[0].device, dtype=local[0].dtype)
        for ci, o in enumerate(local):
            stack[:, ci, :o.shape[1]] = o
        res = shard.gather_uneven(dist, stack, [len(r) for r in shards], world)   # (num_repeats, clips, t_max, 32) on every rank
        if rank == 0:
            tables = res.cpu().numpy()
            for ci, (pid, stem, frames) in enumerate(clips):
                write_clip(args, pid, stem, frames, tables[:, ci])


def _rank_main(args):
    from said_amd import shard
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    torch.cuda.set_device(rank)
    dist = shard.init_process_group("nccl", rank, world, torch.device("cuda", rank))   # "nccl" IS RCCL on ROCm
    try:
        run_rank(args, rank, world, dist)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def main(argv=None) -> None:
    args = build_parser().parse_args(argv)
    if args.gpus > 1:
        from said_amd import shard
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"test_inference.py --gpus {args.gpus}: only {have} MI355X device(s) visible to this process")
        if "WORLD_SIZE" in os.environ:          # started under torch.distributed.run
            _rank_main(args)
        else:
            shard.spawn(_rank_main, (args,), args.gpus)
        return
    run_rank(args, 0, 1, None)


if __name__ == "__main__":
    main()
