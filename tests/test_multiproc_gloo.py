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


# ----------------------------------------------------------------------------- data-parallel training protocol (parallel.py)
def _dp_worker(rank, world, port, q):
    """Two ranks, half a batch each (CPU tensors, gloo): the Sync-CBN moment all-reduce + finalise must reproduce the
    full-batch statistics / gradients, and GradAllReduce must average the parameter gradients.  The arithmetic between
    the collectives is the CPU oracle here (there is no CPU product path); the same hooks drive the HIP kernels on the GPU
    (tests/test_gpu_dp.py)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("i-dccrn-vae_amd.parallel")
    from oracle import idccrn_oracle as O
    g = torch.Generator().manual_seed(5)
    B, C, F, T = 4, 6, 9, 7
    y = (torch.randn(B, C, F, T, 2, generator=g) * 1.5 + 0.3).double()
    w = torch.randn(C, generator=g).double()
    mine = par.shard(y, rank, world)
    # forward: per-rank moment sums [C][5] (sum r, i, rr, ii, ri) -> all-reduce -> global statistics
    r, im = mine[..., 0], mine[..., 1]
    sums = torch.stack([r.sum((0, 2, 3)), im.sum((0, 2, 3)), (r * r).sum((0, 2, 3)), (im * im).sum((0, 2, 3)),
                        (r * im).sum((0, 2, 3))], dim=1).contiguous()
    factor = par.sync_moments(sums)
    n = mine.shape[0] * F * T * factor
    mu_r, mu_i = sums[:, 0] / n, sums[:, 1] / n
    Vrr = sums[:, 2] / n - mu_r ** 2 + 1e-5
    Vii = sums[:, 3] / n - mu_i ** 2 + 1e-5
    Vri = sums[:, 4] / n - mu_r * mu_i
    want = O.cbn_batch_stats(y)
    err = max(float((a - b.reshape(-1)).abs().max()) for a, b in zip((mu_r, mu_i, Vrr, Vri, Vii), want))
    # gradients: a parameter used by every sample; the local loss is the mean over the SHARD, as in the trainers
    lin = torch.nn.Linear(3, 2).double()
    with torch.no_grad():
        lin.weight.copy_(torch.arange(6.0).reshape(2, 3) / 7)
        lin.bias.fill_(0.1)
    unused = torch.nn.Parameter(torch.zeros(2).double())          # no gradient on any rank
    x = torch.randn(8, 3, generator=g).double()
    t = torch.randn(8, 2, generator=g).double()
    xs, ts = par.shard(x, rank, world), par.shard(t, rank, world)
    ((lin(xs) - ts) ** 2).mean().backward()
    par.GradAllReduce([lin.weight, lin.bias, unused], bucket_bytes=32).reduce()      # several small buckets
    lin2 = torch.nn.Linear(3, 2).double()
    lin2.load_state_dict(lin.state_dict())
    ((lin2(x) - t) ** 2).mean().backward()
    gerr = max(float((lin.weight.grad - lin2.weight.grad).abs().max()), float((lin.bias.grad - lin2.bias.grad).abs().max()))
    q.put((rank, factor, err, gerr, 0.0 if unused.grad is None else 1.0))      # no gradient anywhere: stays None, as on one device
    dist.destroy_process_group()


def test_two_rank_sync_bn_and_grad_average():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, factor, err, gerr, unused in out:
        assert factor == 2
        assert err < 1e-12, err                  # global statistics from the all-reduced sums == full-batch statistics
        assert gerr < 1e-12, gerr                # averaged shard gradients == full-batch gradient
        assert unused == 0.0


def test_parallel_single_process_is_a_no_op():
    par = importlib.import_module("i-dccrn-vae_amd.parallel")
    ops = importlib.import_module("i-dccrn-vae_amd").ops
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    par.GradAllReduce([p]).reduce()
    assert torch.equal(p.grad, torch.full((3,), 2.0))
    par.enable_sync_bn()
    assert ops.BN_SYNC is None
    s = torch.ones(2, 5, dtype=torch.float64)
    assert par.sync_moments(s) == 1 and torch.equal(s, torch.ones(2, 5, dtype=torch.float64))
    assert par.shard(torch.arange(10), 1, 2).tolist() == [5, 6, 7, 8, 9]
