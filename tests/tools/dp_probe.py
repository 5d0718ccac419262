"""Diagnostic for tests/test_gpu_dp.py: per conv block, per utterance norms of the block output (forward) and of the gradient
arriving at it (backward), two gloo ranks on one GPU against the single-process full batch.  python tests/tools/dp_probe.py"""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import idccrn_oracle as O  # noqa: E402

NFFT, HOP, WIN = 512, 100, 400
SKIP = [0, 1, 2, 3, 4, 5]


def step(rank, world):
    pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
    nl = importlib.import_module("i-dccrn-vae_amd.model.nsvae_loss")
    par = importlib.import_module("i-dccrn-vae_amd.parallel")
    AG = importlib.import_module("i-dccrn-vae_amd.autograd")
    rec = []
    of, ob = AG.ConvBlockFn.forward, AG.ConvBlockFn.backward

    def per_utt(buf, geom):
        pl = AG._mk(buf, geom)
        t = pl.tensor5().double()                      # [B, C, F, T, 2]
        return [float(v) for v in t.flatten(1).norm(dim=1).cpu()]

    def fwd(ctx, meta, *a):
        out = of(ctx, meta, *a)
        rec.append(("fwd", meta["conv"].out_channel, meta["conv"]._transposed, per_utt(out, ctx.zgeom)))
        return out

    def bwd(ctx, dz):
        rec.append(("bwd_in", ctx.meta["conv"].out_channel, ctx.meta["conv"]._transposed, per_utt(dz.contiguous(), ctx.zgeom)))
        r = ob(ctx, dz)
        if r[1] is not None:
            rec.append(("bwd_dx", ctx.meta["conv"].out_channel, ctx.meta["conv"]._transposed, per_utt(r[1], ctx.meta["x"])))
        return r
    AG.ConvBlockFn.forward, AG.ConvBlockFn.backward = staticmethod(fwd), staticmethod(bwd)
    ops = importlib.import_module("i-dccrn-vae_amd").ops
    ocb = ops.cbn_bwd

    def cbn_bwd(dz, y, fold, moments, bn, slope, count, want_image=False):
        r = ocb(dz, y, fold, moments, bn, slope, count, want_image)
        dy = r[0][0] if want_image else r[0]
        t = lambda pl: [float(v) for v in pl.tensor5().double().flatten(1).norm(dim=1).cpu()]
        rec.append(("t_y", y.C, True, y.tensor5().cpu().numpy().copy()))
        rec.append(("t_dz", y.C, True, dz.tensor5().cpu().numpy().copy()))
        rec.append(("t_dy", y.C, True, dy.tensor5().cpu().numpy().copy()))
        rec.append(("bn_y", y.C, True, t(y)))
        rec.append(("bn_dy", y.C, True, t(dy)))
        rec.append(("bn_par", y.C, True, [float(fold.double().norm()), float(moments.double().norm()), float(r[1].double().norm()),
                                         float(r[2].double().norm())]))
        return r
    ops.cbn_bwd = cbn_bwd
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
    return rec


def worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, step(rank, world)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join()
    full = step(0, 1)
    for k, (kind, cout, tr, vals) in enumerate(full):
        a, b = out[0][k], out[1][k]
        assert a[0] == kind and a[1] == cout
        if kind.startswith("t_"):
            dp = np.concatenate([a[3], b[3]], axis=0).astype("float64") * (0.5 if kind != "t_y" else 1.0)
            d = np.abs(dp - vals.astype("float64"))
            idx = np.unravel_index(np.argmax(d), d.shape)
            big = int((d > 1e-4 * np.abs(vals).max()).sum())
            print(f"{k:3d} {kind:7s} cout={cout:3d} max abs diff {d.max():.3e} at {idx} (value {vals[idx]:.5g}, scale {np.abs(vals).max():.3g}); elements off by > 1e-4 of the scale: {big} of {d.size}")
            continue
        dp_vals = a[3] + b[3]
        scale = 0.5 if kind in ("bwd_in", "bwd_dx", "bn_dy") else 1.0     # rank losses are means over 2 utterances, the full loss over 4
        if kind == "bn_par":
            print(f"{k:3d} {kind:7s} cout={cout:3d} rank0 {['%.7g' % v for v in a[3]]} rank1 {['%.7g' % v for v in b[3]]} full {['%.7g' % v for v in vals]}")
            continue
        rel = [abs(x * scale - y) / (abs(y) + 1e-30) for x, y in zip(dp_vals, vals)]
        print(f"{k:3d} {kind:7s} cout={cout:3d} tr={int(tr)} max rel diff {max(rel):.2e}   {['%.6g' % v for v in vals]}")
