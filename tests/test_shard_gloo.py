"""world_size-2 gloo test of the multi-GPU sharding logic used by bench.py: contiguous clip
partition, no data-path collective, one all-gather of the (B_local, T, 32) results.  Runs on CPU; the per-rank
"path" here is a deterministic stand-in keyed by the global clip index (the HIP path itself needs a GPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B_local, T, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clips = [rank * B_local + i for i in range(B_local)]                 # contiguous shard
    local = torch.stack([torch.full((T, 32), float(c)) + torch.arange(32).float() / 100 for c in clips])
    gathered = torch.empty(world * B_local, T, 32)
    dist.all_gather_into_tensor(gathered, local)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                            # bench.py's max-over-ranks timing
    dist.barrier()
    if rank == 0:
        out_q.put((gathered, float(t)))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world, B_local, T = 2, 3, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B_local, T, q)) for r in range(world)]
    for p in procs: p.start()
    gathered, tmax = q.get(timeout=120)
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert gathered.shape == (world * B_local, T, 32)
    for c in range(world * B_local):                                     # global clip order preserved
        assert torch.equal(gathered[c, 0], torch.full((32,), float(c)) + torch.arange(32).float() / 100)
    assert tmax == 2.0
