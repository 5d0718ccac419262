"""The RCCL branch of parallel.py on the one GPU of the test box.

A process group of ONE rank over backend "nccl" (= RCCL on ROCm) is created in a spawned child -- RCCL before any other GPU
work of that process -- and ``parallel.FORCE_COLLECTIVES`` makes every collective of the data-parallel step be ISSUED although
the group has one rank (sum / average over one rank = identity):

  * ``all_reduce_sum_`` on an fp64 [C][5] device tensor (the Sync-CBN moment sums; parallel.py all_reduce_sum_),
  * ``GradAllReduce.reduce()`` down the ``ReduceOp.AVG`` branch on the device bucket (multi-tensor gather / scatter kernels),
  * one supervised DCCRN-CL train step (train.py:233-243 call order) with ``enable_sync_bn()`` + bucket + the HIP Adam reading the
    all-reduced bucket, against the same step without a process group.

What this does NOT show: any N > 1 behaviour (two ranks on one device are refused by RCCL; the 2 / 4 / 8-GPU curve is the
driver's to measure).  It shows that the dtypes, ops and device buffers the step hands to RCCL are accepted and leave the
results unchanged."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _step(pm, nl, par, optim, use_group: bool):
    np_ = O.net_params(True, 4)
    m = pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None)
    m.load_state_dict(O.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 17))
    m = m.cuda()
    g = torch.Generator().manual_seed(4)
    noisy = (torch.randn(4, 1600, generator=g) * 0.1).cuda()
    clean = (noisy.cpu() + torch.randn(4, 1600, generator=g) * 0.05).cuda()
    params = [q for q in m.parameters() if q.requires_grad]
    opt = optim.Adam(params, lr=1e-3, weight_decay=1e-3)
    par.FORCE_COLLECTIVES = use_group
    if use_group:
        par.enable_sync_bn()
        assert importlib.import_module("i-dccrn-vae_amd.ops").BN_SYNC is not None
    else:
        par.disable_sync_bn()
    red = par.GradAllReduce(params)
    with torch.enable_grad():
        est, pred = m(noisy, train=True)
        loss = nl.ete_train_se_loss([0.2, 0.1, 1.0]).final_ete_loss(pred, m.stft(clean), clean, est)[0]
        loss.backward()
    grads = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in m.named_parameters()}
    if use_group:
        red.reduce(into_grads=True)                       # AVG all-reduce + scatter back: .grad must be unchanged
        for k, v in m.named_parameters():
            if grads[k] is not None:
                assert torch.equal(v.grad, grads[k]), k
        red.reduce(into_grads=False)                      # ... and the hand-over form the optimiser reads
        opt.step(grad_bucket=red.bucket())
    else:
        opt.step()
    torch.cuda.synchronize()
    par.disable_sync_bn()
    par.FORCE_COLLECTIVES = False
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items() if v is not None}
    return float(loss.detach()), {k: (None if v is None else v.cpu().numpy()) for k, v in grads.items()}, sd


def _child(port, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        info = {"backend": dist.get_backend(), "world": dist.get_world_size()}
        pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
        nl = importlib.import_module("i-dccrn-vae_amd.model.nsvae_loss")
        par = importlib.import_module("i-dccrn-vae_amd.parallel")
        optim = importlib.import_module("i-dccrn-vae_amd.optim")
        # 1. the Sync-CBN moment all-reduce: fp64 [C][5] device tensor, SUM
        par.FORCE_COLLECTIVES = True
        s = torch.arange(64 * 5, dtype=torch.float64, device="cuda").reshape(64, 5) * 1.25 + 0.1
        s0 = s.clone()
        assert par.sync_moments(s) == 1
        torch.cuda.synchronize()
        info["moments_equal"] = bool(torch.equal(s, s0))
        # 2. the gradient bucket: odd sizes, a parameter without gradient, AVG on the device buffer
        ps = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (1, 7, 1024, 33, 5 * 2 * 16 * 8)]
        for k, q_ in enumerate(ps):
            if k != 1:
                q_.grad = torch.randn_like(q_)
        g0 = [None if q_.grad is None else q_.grad.clone() for q_ in ps]
        red = par.GradAllReduce(ps)
        red.reduce()
        torch.cuda.synchronize()
        info["bucket_equal"] = all(torch.equal(q_.grad, g) for q_, g in zip(ps, g0) if g is not None)
        info["bucket_none_kept"] = ps[1].grad is None
        par.FORCE_COLLECTIVES = False
        # 3. the train step with and without the group
        with_group = _step(pm, nl, par, optim, True)
        without = _step(pm, nl, par, optim, False)
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", info, with_group, without))
    except BaseException:
        import traceback
        q.put(("error", traceback.format_exc()))
        raise


def test_rccl_one_rank_executes_the_data_parallel_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_child, args=(_free_port(), q))
    pr.start()
    r = q.get(timeout=600)
    pr.join(timeout=120)
    if r[0] == "error":
        pytest.fail("child failed:\n" + r[1])
    assert pr.exitcode == 0
    _, info, a, b = r
    print("rccl one-rank:", info)
    assert info["backend"] == "nccl" and info["world"] == 1
    assert info["moments_equal"] and info["bucket_equal"] and info["bucket_none_kept"]
    # the step behind RCCL collectives == the step without a process group.  The train-mode moment sums are accumulated
    # with double atomics whose order varies from run to run (last digits), so "equal" is 1e-6 relative, not bitwise;
    # bitwise equality is reported.
    assert abs(a[0] - b[0]) <= 1e-6 * max(1.0, abs(b[0])), (a[0], b[0])
    worst, bit = 0.0, True
    for k, gb in b[1].items():
        ga = a[1][k]
        assert (ga is None) == (gb is None), k
        if gb is None:
            continue
        bit = bit and np.array_equal(ga, gb)
        e = float(np.linalg.norm(ga.astype("float64") - gb)) / (float(np.linalg.norm(gb.astype("float64"))) + 1e-30)
        if k.endswith("conv_re.bias") or k.endswith("conv_im.bias") or k.endswith("tconv_re.bias") or k.endswith("tconv_im.bias"):
            continue                                     # bias in front of a batch norm: exactly-zero gradient, noise only
        worst = max(worst, e)
        assert e < 1e-5, (k, e)
    for k, vb in b[2].items():                           # parameters after Adam (bucket hand-over vs .grad) and BN buffers
        va = a[2][k]
        e = float(np.linalg.norm(va.astype("float64") - vb)) / (float(np.linalg.norm(vb.astype("float64"))) + 1e-30)
        assert e < 1e-4, (k, e)                          # (the first Adam step is sign-like: an element with |g| ~ 1e-8 may differ)
    print(f"rccl one-rank step: worst gradient deviation {worst:.2e}, gradients bitwise equal: {bit}")


def test_bench_distributed_path_with_one_rccl_rank():
    """bench.py exactly as the driver launches an N-rank run -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` -- with N = 1 and IDV_BENCH_FORCE_DIST=1 / IDV_DP_FORCE=1: the
    rank creates the RCCL process group before any other GPU work, runs a data-parallel train workload (Sync-CBN all-reduces,
    gradient bucket through RCCL, Adam from the bucket), the barriers and the timing all-reduce, and prints the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IDV_BENCH_FORCE_DIST="1", IDV_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "dccrn_cl_train",
           "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["rccl_ranks"] == 1 and d["backend"].startswith("nccl") and d["n_gpus"] == 1
    assert d["value"] > 0 and d["steps"] == 2 and d["scaling"] == "weak"
    assert "data-parallel" in d["config"]["parallelism"]
