"""Multi-tensor train-step kernels (csrc/bucket.hip) against stock torch: the gradient bucket's gather / scatter and the
one-launch Adam against ``torch.optim.Adam`` -- the optimiser every reference trainer constructs
(``optim.Adam(model.parameters(), lr=learning_rate, weight_decay=0.001)``: supervised_dccrn/train.py:109,
i_dccrn_vae/nsvae_dccrn/train_nsvae.py:200) -- over several steps, with ``ReduceLROnPlateau`` changing the rate and a
state_dict round trip in both directions (checkpoint compatibility)."""
import copy
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (3, 5), (257,), (64, 32, 5, 2), (2,), (1024, 1280), (7, 1, 1)]


@pytest.fixture(scope="module")
def optim():
    return importlib.import_module("i-dccrn-vae_amd.optim")


def _params(seed, skip_grad=()):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).cuda()) for s in SHAPES]
    return ps


def _set_grads(ps, seed, skip=()):
    g = torch.Generator().manual_seed(seed)
    for k, q in enumerate(ps):
        gr = (torch.randn(*q.shape, generator=g) * (0.5 if k % 2 else 2e-3)).cuda()
        q.grad = None if k in skip else gr


def test_bucket_gather_scatter(optim):
    ps = _params(1)
    _set_grads(ps, 2, skip=(4,))
    # an unaligned tensor (a view that starts 4 bytes into an allocation) takes the scalar path
    base = torch.randn(1001, device="cuda")
    ps.append(torch.nn.Parameter(torch.zeros(1000, device="cuda")))
    ps[-1].grad = base[1:]
    assert ps[-1].grad.data_ptr() % 16 != 0
    tab = optim.TensorTable([q.numel() for q in ps], ps[0].device)
    flat = tab.flat(zero=False).fill_(float("nan"))
    grads = [q.grad for q in ps]
    optim.bucket_gather(tab, grads, flat)
    for q, o, n in zip(ps, tab.offsets, tab.numels):
        want = torch.zeros(n, device="cuda") if q.grad is None else q.grad.reshape(-1)
        assert torch.equal(flat[o:o + n], want)
        pad = (n + 3) // 4 * 4 - n
        assert pad == 0 or bool((flat[o + n:o + n + pad] == 0).all())
    assert tab.total == flat.numel() and not torch.isnan(flat).any()
    outs = [torch.full_like(q, 7.0) for q in ps]
    outs[-1] = torch.full((1001,), 7.0, device="cuda")[1:]
    optim.bucket_scatter(tab, outs, flat, 0.5)
    for q, out in zip(ps, outs):
        want = torch.zeros_like(q) if q.grad is None else 0.5 * q.grad
        assert torch.equal(out.reshape(q.shape), want)
    # absent destinations are skipped, the others still written
    outs2 = [None if k == 2 else torch.zeros_like(q) for k, q in enumerate(ps)]
    optim.bucket_scatter(tab, outs2, flat, 1.0)
    assert torch.equal(outs2[3], ps[3].grad)


@pytest.mark.parametrize("wd", [0.0, 1e-3])
def test_adam_matches_torch_adam(optim, wd):
    a = _params(3)
    b = [torch.nn.Parameter(q.detach().clone()) for q in a]
    oa = optim.Adam(a, lr=1e-3, weight_decay=wd)
    ob = torch.optim.Adam(b, lr=1e-3, weight_decay=wd)
    sa = torch.optim.lr_scheduler.ReduceLROnPlateau(oa, 'min', factor=0.5, patience=0)
    sb = torch.optim.lr_scheduler.ReduceLROnPlateau(ob, 'min', factor=0.5, patience=0)
    for step in range(6):
        skip = (4,)                                       # one parameter never receives a gradient: skipped, no state
        _set_grads(a, 10 + step, skip)
        _set_grads(b, 10 + step, skip)
        v0 = [q._version for q in a]
        oa.step()
        ob.step()
        assert all((q._version > v) == (k != 4) for q, v, k in zip(a, v0, range(len(a))))     # pack caches key on _version
        sa.step(1.0 + step)                               # a rising "validation loss": the rate halves
        sb.step(1.0 + step)
        assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
    assert oa.param_groups[0]["lr"] < 1e-3
    for k, (q, r) in enumerate(zip(a, b)):
        # fp32 round-off of a different evaluation order (fused multiply-adds): 1e-6 relative to the parameter scale
        assert float((q - r).abs().max()) <= 2e-6 * float(r.abs().max()), (k, float((q - r).abs().max()))
        if k == 4:
            assert torch.equal(q, r) and q not in oa.state
        else:
            assert float((oa.state[q]["exp_avg"] - ob.state[r]["exp_avg"]).abs().max()) <= 1e-6 * float(ob.state[r]["exp_avg"].abs().max())
            assert float((oa.state[q]["exp_avg_sq"] - ob.state[r]["exp_avg_sq"]).abs().max()) <= 1e-6 * float(ob.state[r]["exp_avg_sq"].abs().max())
            assert float(oa.state[q]["step"]) == float(ob.state[r]["step"]) == 6.0
    # state_dict round trips: ours -> torch's, torch's -> ours, then one more step on each side
    sd_ours, sd_torch = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    assert sd_ours["state"].keys() == sd_torch["state"].keys()
    assert set(next(iter(sd_ours["state"].values())).keys()) == {"step", "exp_avg", "exp_avg_sq"}
    a2 = [torch.nn.Parameter(q.detach().clone()) for q in b]
    b2 = [torch.nn.Parameter(q.detach().clone()) for q in b]
    oa2 = optim.Adam(a2, lr=1e-3, weight_decay=wd)
    ob2 = torch.optim.Adam(b2, lr=1e-3, weight_decay=wd)
    oa2.load_state_dict(sd_torch)
    ob2.load_state_dict(sd_ours)
    _set_grads(a2, 99, (4,))
    _set_grads(b2, 99, (4,))
    oa2.step()
    ob2.step()
    for k, (q, r) in enumerate(zip(a2, b2)):
        assert float((q - r).abs().max()) <= 2e-6 * float(r.abs().max()), k
    assert float(oa2.state[a2[0]]["step"]) == 7.0


def test_adam_from_bucket(optim):
    """Gradients read straight from a bucket (what GradAllReduce hands over) == gradients read from .grad, scaled."""
    a = _params(5)
    b = [torch.nn.Parameter(q.detach().clone()) for q in a]
    _set_grads(a, 6)
    _set_grads(b, 6)
    oa, ob = optim.Adam(a, lr=3e-4, weight_decay=1e-3), optim.Adam(b, lr=3e-4, weight_decay=1e-3)
    tab = optim.TensorTable([q.numel() for q in a], a[0].device)
    flat = tab.flat()
    optim.bucket_gather(tab, [q.grad for q in a], flat)
    flat.mul_(4.0)                                         # a SUM over four ranks ...
    oa.step(grad_bucket=(tab, flat, 0.25, [True] * len(a)))     # ... averaged by the scale
    ob.step()
    for q, r in zip(a, b):
        assert torch.equal(q, r)
    # a parameter marked absent is left alone (value, moments), the others step on
    keep = a[2].detach().clone()
    flat.zero_()
    oa.step(grad_bucket=(tab, flat, 1.0, [k != 2 for k in range(len(a))]))
    assert torch.equal(a[2], keep) and not torch.equal(a[3], b[3])
