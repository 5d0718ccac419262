"""CPU, world_size 2, gloo: the N > 1 reduction bench.py uses (max time over ranks, summed utterances) and
the batch sharding helper."""
import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dt = importlib.import_module("i-dccrn-vae_amd.utils.dist_timing")
    elapsed = 2.0 if rank == 0 else 4.0                 # rank 1 is the slow one
    value, tmax = dt.job_throughput(elapsed, 64.0)
    lo, hi = dt.shard_batch(9, rank, world)
    dist.barrier()
    q.put((rank, value, tmax, lo, hi))
    dist.destroy_process_group()


def test_two_rank_throughput_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, value, tmax, lo, hi in out:
        assert tmax == 4.0 and value == 128.0 / 4.0       # all ranks' utterances / slowest rank
    assert (out[0][3], out[0][4], out[1][3], out[1][4]) == (0, 5, 5, 9)


def test_single_process_path():
    dt = importlib.import_module("i-dccrn-vae_amd.utils.dist_timing")
    assert dt.job_throughput(2.0, 64.0) == (32.0, 2.0)
    assert [dt.shard_batch(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
