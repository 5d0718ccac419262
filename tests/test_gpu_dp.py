"""Data-parallel training on the HIP path (BASELINE configs 4 / 5): two ranks with half a batch each
(Sync-CBN moment all-reduce in forward and backward + gradient averaging, parallel.py) must reproduce the single-process
step on the full batch: same loss, same running batch-norm buffers, same parameter gradients.  The ranks share the one
GPU of the test box and talk over gloo (device tensors staged through the host); on an 8-GPU node the same code runs over
RCCL (bench.py --gpus N --workload dccrn_cl_train | nsvae_train | twophase_train)."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import idccrn_oracle as O

pytestmark = pytest.mark.gpu
NFFT, HOP, WIN = 512, 100, 400
SKIP = [0, 1, 2, 3, 4, 5]
KEYS = ("std_DCCRN.encoders.0.conv.conv_re.weight", "std_DCCRN.encoders.3.bn.gamma_ri", "std_DCCRN.encoders.5.prelu.weight",
        "std_DCCRN.lstms.0.lstm_im.weight_hh_l1", "std_DCCRN.dense.linear_read.weight", "std_DCCRN.decoders.1.transconv.tconv_im.weight",
        "std_DCCRN.decoders.4.bn.beta_r", "std_DCCRN.decoders.5.transconv.tconv_re.weight", "std_DCCRN.decoders.2.prelu.weight")
BUFS = ("std_DCCRN.encoders.2.bn.Vri", "std_DCCRN.decoders.3.bn.running_mean_real", "std_DCCRN.decoders.5.bn.Vii")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train_step(rank, world):
    pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
    nl = importlib.import_module("i-dccrn-vae_amd.model.nsvae_loss")
    par = importlib.import_module("i-dccrn-vae_amd.parallel")
    np_ = O.net_params(True, 4)
    m = pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None)
    m.load_state_dict(O.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 17))
    m = m.cuda()
    g = torch.Generator().manual_seed(4)
    noisy = torch.randn(4, 1600, generator=g) * 0.1
    clean = noisy + torch.randn(4, 1600, generator=g) * 0.05
    noisy, clean = par.shard(noisy, rank, world).cuda(), par.shard(clean, rank, world).cuda()
    par.enable_sync_bn()
    red = par.GradAllReduce(m.parameters())
    with torch.enable_grad():
        est, pred = m(noisy, train=True)
        loss = nl.ete_train_se_loss([0.2, 0.1, 1.0]).final_ete_loss(pred, m.stft(clean), clean, est)[0]
        loss.backward()
    red.reduce()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    sd = m.state_dict()
    # numpy (pickled by value): torch tensors would travel through shared-memory handles that die with the worker
    return (float(loss.detach()), {k: params[k].grad.cpu().numpy() for k in KEYS}, {k: sd[k].cpu().numpy() for k in BUFS},
            {k: float(v.grad.double().norm()) for k, v in params.items() if v.grad is not None})


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _train_step(rank, world))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_rank_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full_loss, full_g, full_b, full_n = _train_step(0, 1)
    # mean of the shard losses == loss of the full batch (equal shards)
    assert abs(0.5 * (out[0][1] + out[1][1]) - full_loss) < 1e-5 * max(1.0, abs(full_loss))
    for rank, _, grads, bufs, norms in out:
        for k in KEYS:
            ref = full_g[k].astype("float64")
            assert float(np.linalg.norm(grads[k].astype("float64") - ref)) <= 1e-5 * float(np.linalg.norm(ref)) + 1e-9, (rank, k)
        for k in BUFS:
            ref = full_b[k].astype("float64")
            assert float(np.linalg.norm(bufs[k].astype("float64") - ref)) <= 1e-5 * float(np.linalg.norm(ref)) + 1e-9, (rank, k)
        for k, v in full_n.items():
            if k.endswith("conv_re.bias") or k.endswith("conv_im.bias"):
                continue                          # bias in front of a batch norm: the true gradient is exactly zero
            assert abs(norms[k] - v) <= 2e-5 * v + 1e-7, (rank, k, norms[k], v)
    # both ranks hold identical (averaged) gradients
    for k in KEYS:
        assert np.array_equal(out[0][2][k], out[1][2][k]), k
