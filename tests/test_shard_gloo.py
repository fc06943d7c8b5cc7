"""world_size-2 gloo tests of the multi-GPU path on CPU.  They drive the SAME code bench.py runs on
RCCL — said_amd.shard.{clip_range, init_process_group, timed_sharded_passes, gather_clips, spawn} — with a CPU
stand-in for the per-rank path (the HIP path itself needs a GPU): contiguous clip partition, no data-path collective,
one all-gather in global clip order, max-over-ranks timing, and bench.py's self-launch of N ranks."""
import json
import os
import subprocess
import sys
import time

import torch
import torch.multiprocessing as mp

from said_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clip_tensor(c, T):
    return torch.full((T, 32), float(c)) + torch.arange(32).float() / 100


def _worker(rank, world, port, B_local, T, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist = shard.init_process_group("gloo", rank, world)
    calls = []

    def path_fn(clips):                      # stand-in keyed by the GLOBAL clip id; rank 1 is the slow rank
        calls.append(list(clips))
        if rank == 1:
            time.sleep(0.05)
        return torch.stack([_clip_tensor(c, T) for c in clips])

    r = shard.timed_sharded_passes(path_fn, rank=rank, world=world, clips_per_rank=B_local, steps=3, warmup=2, dist=dist,
                                   device=torch.device("cpu"))
    out_q.put((rank, r.gathered.numpy().copy(), r.elapsed_s, r.clip_ranges, r.checksum, calls))   # by value: a tensor travels as a file descriptor its (exited) sender must still serve
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world, B_local, T = 2, 3, 7
    res = _run_world(world, B_local, T)
    want = torch.stack([_clip_tensor(c, T) for c in range(world * B_local)])
    for rank, gathered, elapsed, ranges, checksum, calls in res:
        assert torch.equal(torch.from_numpy(gathered), want)                               # global clip order, on every rank
        assert ranges == [[0, 2], [3, 5]]
        assert checksum == float(want.double().sum())
        assert len(calls) == 5 and all(c == list(shard.clip_range(rank, world, B_local)) for c in calls)   # 2 warm-up + exactly 3 timed
    # elapsed is the MAX over ranks: both ranks report the slow rank's time (3 timed passes x 50 ms)
    assert res[0][2] == res[1][2] and res[0][2] >= 0.15


def _run_world(world, B_local, T):
    """Spawns the ranks and collects their results.  A rendezvous that does not come up (the free port taken between probing and binding, a loaded build container
    starting 8 interpreters slowly) is retried once on a fresh port; what the ranks RETURN is asserted by the callers and never retried."""
    import queue
    ctx = mp.get_context("spawn")
    last = None
    for attempt in range(2):
        q = ctx.Queue()
        port = shard.free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, B_local, T, q)) for r in range(world)]
        for p in procs: p.start()
        try:
            res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
            for p in procs: p.join(timeout=120)
            if all(p.exitcode == 0 for p in procs):
                return res
            last = f"exit codes {[p.exitcode for p in procs]}"
        except queue.Empty:
            last = "a rank did not report within 240 s"
        for p in procs:
            if p.is_alive():
                p.kill()
            p.join(timeout=30)
    raise AssertionError(f"gloo world {world} did not complete in two attempts: {last}")


def test_shard_and_gather_world4_and_world8():
    """BASELINE configs[3]'s world size (8) and the 4-GPU point of the scaling curve, on the gloo stand-in: contiguous global
    clip order on every rank, identical max-over-ranks time everywhere, exactly warm-up + K passes per rank."""
    for world, B_local, T in ((4, 2, 5), (8, 2, 3)):
        res = _run_world(world, B_local, T)
        want = torch.stack([_clip_tensor(c, T) for c in range(world * B_local)])
        for rank, gathered, elapsed, ranges, checksum, calls in res:
            assert torch.equal(torch.from_numpy(gathered), want)
            assert ranges == [[r * B_local, (r + 1) * B_local - 1] for r in range(world)]
            assert checksum == float(want.double().sum())
            assert len(calls) == 5 and all(c == list(shard.clip_range(rank, world, B_local)) for c in calls)
        assert len({r[2] for r in res}) == 1 and res[0][2] >= 0.15     # the slow rank's time, reported by all


def test_bench_rank_failure_ends_every_rank_no_hang():
    """One rank raising inside its path must take the whole job down promptly (non-zero exit, no rank left waiting in the
    all-gather): bench.py's self-launch over 4 gloo ranks with the stand-in path failing on rank 2."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--batch", "1", "--steps", "2", "--warmup", "1",
                          "--seconds", "0.5", "--dry_run_gloo", "--dry_run_fail_rank", "2"], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode != 0
    assert "stand-in path failure on rank 2" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]     # no result line from a failed job
    assert time.time() - t0 < 200


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` with fewer than N visible devices fails fast with a clear message instead of spawning ranks that die
    one by one (here: no GPU at all, or a single one)."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("8 GPUs visible")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, timeout=240, env=env)
    assert out.returncode != 0 and "--gpus 8" in out.stderr and "visible" in out.stderr


def test_clip_range_partitions_without_overlap():
    for world, B in [(1, 1), (2, 3), (8, 32)]:
        ids = [c for r in range(world) for c in shard.clip_range(r, world, B)]
        assert ids == list(range(world * B))


def test_bench_self_launches_ranks_dry_run():
    """`python bench.py --gpus 2` without torchrun must spawn its two ranks itself; on CPU this is
    exercised through the gloo dry-run flag, which shares launch, shard, gather and timing code with the RCCL run."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "2", "--warmup", "1",
                          "--seconds", "0.5", "--dry_run_gloo"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                               # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["clip_ranges"] == [[0, 1], [2, 3]] and j["gathered_shape"] == [4, 30, 32] and j["checksum_ok"]


# ---------------------------------------------------------------- round 4: the product entry point, uneven shards
def _item_tensor(i, T):
    g = torch.Generator().manual_seed(1000 + i)      # seeded by the GLOBAL item id, as script/test_inference.py --gpus N seeds a repeat
    return torch.randn(T, 32, generator=g)


def _uneven_worker(rank, world, port, n_items, T, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist = shard.init_process_group("gloo", rank, world)
    seen = []

    def path_fn(ids):
        seen.append(list(ids))
        return torch.stack([_item_tensor(i, T) for i in ids]) if len(ids) else torch.zeros(0, T, 32)

    res = shard.sharded_inference(path_fn, n_items, rank=rank, world=world, dist=dist)
    out_q.put((rank, res.numpy().copy(), seen))          # by value: a tensor travels as a file descriptor its (exited) sender must still serve
    dist.barrier()
    dist.destroy_process_group()


def _run_uneven(world, n_items, T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = shard.free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, n_items, T, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    return res


def test_shard_bounds_uneven():
    assert [len(r) for r in shard.shard_bounds(100, 8)] == [13, 13, 13, 13, 12, 12, 12, 12]
    assert [len(r) for r in shard.shard_bounds(72, 5)] == [15, 15, 14, 14, 14]
    assert [len(r) for r in shard.shard_bounds(3, 8)] == [1, 1, 1, 0, 0, 0, 0, 0]
    for n, w in ((100, 8), (72, 5), (3, 8), (0, 2), (64, 1)):
        flat = [i for r in shard.shard_bounds(n, w) for i in r]
        assert flat == list(range(n))                     # contiguous, complete, in order


def test_sharded_inference_uneven_equals_one_rank():
    """100 items over 8 ranks and 72 over 5 (the reference's num_repeats, script/test_inference.py:90) — shards of different sizes, padded to
    the largest for the ONE all-gather and trimmed after it — give, on every rank, bit for bit the result of a single rank; a world with
    more ranks than items works too (empty shards)."""
    T = 4
    for world, n_items in ((8, 100), (5, 72), (4, 3)):
        want = shard.sharded_inference(lambda ids: torch.stack([_item_tensor(i, T) for i in ids]), n_items, rank=0, world=1)
        res = _run_uneven(world, n_items, T)
        for rank, got, seen in res:
            assert torch.equal(torch.from_numpy(got), want), (world, n_items, rank)
            assert seen == [list(shard.shard_bounds(n_items, world)[rank])]      # ONE call, own shard only


def test_bench_under_torch_distributed_run_two_ranks_gloo():
    """The driver's launch form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 2 ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) — here with the CPU stand-in path over gloo:
    ONE JSON line from rank 0, global clip order, n_gpus = 2."""
    port = shard.free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry_run_gloo", "--seconds", "0.5", "--num_steps", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["checksum_ok"] and j["clip_ranges"] == [[0, 0], [1, 1]]
