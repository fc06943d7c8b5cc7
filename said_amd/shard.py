"""Multi-GPU sharding of the SAiD path: clips are independent, so they are partitioned
contiguously over the ranks (one process per GPU), every rank runs the whole path on its own clips with no
data-path collective, and ONE all-gather (RCCL over xGMI when the backend is "nccl") assembles the
(world * B_local, T, C) result in global clip order.

This module is the single implementation of that logic: ``bench.py`` calls it with the HIP path on "nccl",
``tests/test_shard_gloo.py`` calls the same functions with a CPU stand-in path on "gloo" (world_size 2).
The reference has no counterpart (its inference is single-device: script/inference.py:103-107, 157-158).
"""
from __future__ import annotations

import os
import socket
import time
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch


def clip_range(rank: int, world: int, clips_per_rank: int) -> range:
    """Global clip ids owned by `rank` (contiguous shard; weak scaling: every rank owns `clips_per_rank`)."""
    if not (0 <= rank < world) or clips_per_rank < 1:
        raise ValueError(f"bad shard request: rank {rank} of {world}, {clips_per_rank} clips per rank")
    return range(rank * clips_per_rank, (rank + 1) * clips_per_rank)


def shard_bounds(n_items: int, world: int) -> List[range]:
    """Contiguous partition of `n_items` independent items (clips, or repeats of one clip) over `world` ranks, UNEVEN shards allowed: the
    first `n_items % world` ranks own one item more (100 over 8 -> 13, 13, 13, 13, 12, 12, 12, 12; a rank may own none)."""
    if n_items < 0 or world < 1:
        raise ValueError(f"bad partition request: {n_items} items over {world} ranks")
    q, r = divmod(n_items, world)
    out, lo = [], 0
    for k in range(world):
        hi = lo + q + (1 if k < r else 0)
        out.append(range(lo, hi))
        lo = hi
    return out


def gather_uneven(dist, local: torch.Tensor, counts: List[int], world: int, pad_to: Optional[int] = None) -> torch.Tensor:
    """ONE all-gather of per-rank results with different row counts: every rank pads its (counts[rank], ...) tensor to max(counts) rows,
    `all_gather_into_tensor` assembles (world, max, ...), and the padding rows are trimmed: (sum(counts), ...) in global item order on
    every rank.  (`all_gather_into_tensor` needs equal shapes; a second collective for the sizes is not needed: the partition is a pure
    function of (n_items, world).)"""
    if world == 1 and dist is None:
        return local
    m = max(counts) if pad_to is None else int(pad_to)   # pad_to: a fixed per-rank capacity >= max(counts) (static shapes across calls; the one-rank RCCL test)
    if m < max(counts):
        raise ValueError(f"pad_to {m} is below the largest shard ({max(counts)})")
    if local.shape[0] != m:
        pad = torch.zeros((m - local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        local = torch.cat([local, pad])
    out = torch.empty((world * m,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, local.contiguous())
    if all(c == m for c in counts):
        return out
    return torch.cat([out[k * m: k * m + c] for k, c in enumerate(counts)])


def sharded_inference(path_fn: Callable[[range], torch.Tensor], n_items: int, *, rank: int, world: int, dist=None) -> torch.Tensor:
    """The product's multi-GPU entry point (the reference's batched caller, script/test_inference.py:147-202, is
    single-device): `n_items` independent items are partitioned contiguously over the ranks (shard_bounds), every rank runs
    `path_fn(global item ids of its shard)` -> (n_local, ...) on its own GPU with no data-path collective, and one all-gather returns
    the (n_items, ...) result in global order on every rank.  `path_fn` is given GLOBAL ids so that anything random inside it can be
    seeded per item: the result is then independent of the number of ranks."""
    shards = shard_bounds(n_items, world)
    mine = shards[rank]
    local = path_fn(mine)
    if local.shape[0] != len(mine):
        raise RuntimeError(f"path returned {local.shape[0]} items for a shard of {len(mine)}")
    return gather_uneven(dist, local, [len(r) for r in shards], world)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def init_process_group(backend: str, rank: int, world: int, device: Optional[torch.device] = None):
    """Rendezvous on 127.0.0.1 (the container hostname may not resolve); MASTER_PORT must be set by the launcher."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        raise RuntimeError("MASTER_PORT is not set: launch through torch.distributed.run or said_amd.shard.spawn")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


@dataclass
class ShardedRun:
    elapsed_s: float              # max over ranks of the timed region
    gathered: Optional[torch.Tensor]   # (world * B_local, T, C) in global clip order (every rank holds it)
    clip_ranges: List[List[int]]  # per rank [first, last] global clip ids
    checksum: float               # float64 sum of the gathered tensor (same on every rank)


def gather_clips(dist, local: torch.Tensor, world: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One all-gather of the per-rank (B_local, T, C) results into (world * B_local, T, C): rank r's clips land at
    rows [r * B_local, (r + 1) * B_local) — the contiguous partition of clip_range()."""
    if world == 1 and dist is None:
        return local
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def timed_sharded_passes(path_fn: Callable[[range], torch.Tensor], *, rank: int, world: int, clips_per_rank: int,
                         steps: int, warmup: int, dist=None, device: Optional[torch.device] = None) -> ShardedRun:
    """The bench contract: `warmup` untimed passes, then EXACTLY `steps` passes bracketed by a barrier +
    device synchronisation on both sides; elapsed = MAX over ranks.  One pass = path_fn(own clip ids) +
    (world > 1) the all-gather, both inside the timed region.  With world == 1 and a process group given (`dist`), the
    collectives run as well — a one-rank RCCL all-gather / barrier / all-reduce: the only form in which the "nccl" code
    path can be executed on a single-GPU box."""
    clips = clip_range(rank, world, clips_per_rank)
    on_gpu = device is not None and device.type == "cuda"
    sync = (lambda: torch.cuda.synchronize(device)) if on_gpu else (lambda: None)
    gathered = None

    def one_pass():
        nonlocal gathered
        local = path_fn(clips)
        if local.shape[0] != clips_per_rank:
            raise RuntimeError(f"path returned {local.shape[0]} clips for a shard of {clips_per_rank}")
        if collectives:
            if gathered is None:
                gathered = torch.empty((world * clips_per_rank,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
            gather_clips(dist, local, world, gathered)
        else:
            gathered = local
        return gathered

    collectives = world > 1 or dist is not None
    for _ in range(warmup):
        one_pass()
    if collectives:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    sync()
    if collectives:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if collectives:
        tt = torch.tensor([elapsed], device=device if on_gpu else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    checksum = float(gathered.double().sum().item()) if gathered is not None else 0.0
    ranges = [[clip_range(r, world, clips_per_rank)[0], clip_range(r, world, clips_per_rank)[-1]] for r in range(world)]
    return ShardedRun(elapsed_s=elapsed, gathered=gathered, clip_ranges=ranges, checksum=checksum)


def _spawn_entry(rank: int, world: int, port: int, target, args):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"] = str(world)
    os.environ["RANK"] = str(rank)
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    target(*args)


def spawn(target, args: tuple, world: int) -> None:
    """Self-launch `world` ranks on this node (one per GPU): what `torch.distributed.run --nproc-per-node` would do,
    for callers that were started as a plain `python bench.py --gpus N`."""
    import torch.multiprocessing as mp
    mp.spawn(_spawn_entry, args=(world, free_port(), target, args), nprocs=world, join=True)
