"""Gradient parity on the GPU: the HIP backward kernels behind the torch.autograd.Function classes (autograd.py) against
(a) torch.autograd run through the CPU oracle in float64 on the same seeded inputs, and (b) gradient fixtures produced by
the REAL reference's ``loss.backward()`` / ``Adam.step()`` (tests/golden/make_golden.py ``grads``).  Tolerance: 1e-3
relative on every gradient tensor; most operators are asserted at 2e-4."""
import importlib

import numpy as np
import pytest
import torch

from conftest import relerr
from oracle import idccrn_oracle as O

pytestmark = pytest.mark.gpu
NFFT, HOP, WIN = 512, 100, 400
SKIP = [0, 1, 2, 3, 4, 5]
GTOL = 2e-4


@pytest.fixture(scope="module")
def ops(amd):
    return amd.ops


@pytest.fixture(scope="module")
def cp():
    return importlib.import_module("i-dccrn-vae_amd.model.complex_progress")


@pytest.fixture(scope="module")
def pm():
    return importlib.import_module("i-dccrn-vae_amd.model.pvae_module")


@pytest.fixture(scope="module")
def losses():
    return (importlib.import_module("i-dccrn-vae_amd.model.nsvae_loss"),
            importlib.import_module("i-dccrn-vae_amd.model.pretrain_pvaes_loss"),
            importlib.import_module("i-dccrn-vae_amd.model.sisnr_loss"))


def T_(a):
    return torch.from_numpy(np.asarray(a))


def rnd(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


def leaf64(t):
    return t.double().clone().requires_grad_(True)


def check(name, got, want, tol=GTOL, atol=0.0):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = float((got - want).norm())
    ref = float(want.norm())
    assert err <= tol * ref + atol, f"{name}: |diff| {err:.3e} vs |ref| {ref:.3e} (rel {err / (ref + 1e-30):.2e})"


# ----------------------------------------------------------------------------- conv blocks
@pytest.mark.parametrize("transposed,cin,cout,F,T,B,skip_c,bn", [
    (False, 4, 8, 17, 9, 2, 0, True), (False, 1, 16, 33, 21, 3, 0, True), (False, 32, 64, 17, 40, 2, 0, True),
    (True, 6, 4, 9, 9, 2, 0, True), (True, 8, 8, 9, 33, 3, 8, True), (True, 16, 1, 17, 20, 2, 16, True),
    (True, 64, 32, 5, 70, 2, 64, True), (False, 4, 8, 17, 9, 2, 0, False), (True, 8, 4, 9, 12, 2, 4, False),
    # real widths: several 128-plane S tiles and 32-plane L tiles of the weight-gradient kernel, ragged column tiles
    (True, 256, 128, 5, 37, 2, 256, True), (False, 128, 256, 9, 50, 3, 0, True), (True, 32, 1, 33, 41, 2, 32, True),
])
def test_conv_block_grads(ops, pm, cp, transposed, cin, cout, F, T, B, skip_c, bn):
    """conv / transposed conv (+ skip concat) + train-mode ComplexBatchNormal + PReLU: every gradient vs oracle autograd."""
    _conv_block_grads(ops, pm, transposed, cin, cout, F, T, B, skip_c, bn, 2e-5, GTOL)


@pytest.mark.parametrize("transposed,cin,cout,F,T,B,skip_c,bn", [
    (False, 32, 64, 17, 40, 2, 0, True), (True, 64, 32, 5, 70, 2, 64, True), (True, 256, 128, 5, 37, 2, 256, True),
    (False, 128, 256, 9, 50, 3, 0, True), (True, 64, 64, 9, 33, 2, 64, False),
    # column counts B * (T + 1) that are and are not multiples of 4 (one masked float4 per row at the boundary), several
    # S / L plane tiles, a ragged last column tile, one and many frequency rows
    (True, 64, 32, 5, 39, 2, 64, True), (False, 32, 64, 17, 39, 2, 0, True), (True, 256, 128, 5, 37, 2, 256, True),
    (False, 64, 128, 9, 47, 4, 0, False), (True, 128, 16, 3, 15, 2, 0, True),
])
def test_conv_block_grads_bf16x3(ops, pm, cp, transposed, cin, cout, F, T, B, skip_c, bn):
    """The same blocks in bf16x3 training mode (forward + data gradient on the split-bf16 MFMA kernels, weight gradient on
    the fp32 MFMA): 1e-3 on every gradient (measured ~1e-5)."""
    keep = ops.PRECISION
    ops.set_precision("bf16x3")
    try:
        _conv_block_grads(ops, pm, transposed, cin, cout, F, T, B, skip_c, bn, 1e-4, 1e-3)
    finally:
        ops.set_precision(keep)


def test_wgrad_bf16_offsets_beyond_2_gib(ops):
    """The split-bf16 weight-gradient kernel addresses its operands with 32-bit buffer offsets (out-of-range marker
    0xffffff00): a 2.6 GB activation (the first blocks of the CVAE decoder at B x num_samples = 160) must take the kernel, not
    its fp32 fallback, and agree with the exact-fp32 kernel."""
    cx, cout, F, T, B = 32, 32, 129, 639, 125
    g = torch.Generator(device="cuda").manual_seed(7)
    x = ops.Planar.empty(cx, F, B, T, T + 1, "cuda")
    dy = ops.Planar.empty(cout, (F - 1) // 2 + 1, B, T, T + 1, "cuda")
    for t in (x, dy):
        t.buf.normal_(generator=g)
        pl = t.planes()
        pl[..., 0] = 0.0                                   # guard columns
    assert x.C * 2 * x.F * x.Jp * 4 > 2 ** 31 + 2 ** 28            # 2.6 GB of activation planes
    outs = []
    keep = ops.PRECISION
    try:
        for prec in ("fp32", "bf16x3"):
            ops.set_precision(prec)
            dwr = torch.zeros(cout, cx, 5, 2, device="cuda")
            dwi = torch.zeros_like(dwr)
            ops.cconv_wgrad(x, 0, dy, cout, cx, False, True, dwr, dwi)
            torch.cuda.synchronize()
            outs.append((dwr.cpu().double(), dwi.cpu().double()))
    finally:
        ops.set_precision(keep)
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).norm() / a.norm()) < 1e-4


@pytest.mark.parametrize("transposed,cin,cout,F,T,B,skip_c,bn", [
    (False, 4, 8, 17, 9, 2, 0, True), (False, 32, 64, 17, 40, 2, 0, True), (True, 8, 4, 9, 12, 2, 4, True),
    (True, 64, 32, 5, 33, 2, 64, True), (False, 8, 16, 9, 21, 3, 0, False), (True, 16, 8, 5, 19, 2, 0, False),
])
def test_conv_block_grads_noncausal(ops, pm, cp, transposed, cin, cout, F, T, B, skip_c, bn):
    """The NON-causal blocks of model/net_config.py (padding (2, 0): the conv drops a frame, the transposed conv adds one;
    reference model/complex_progress.py:24-36, :253-279) through the same autograd.Function: forward, data gradient (the adjoint
    with the other tap alignment), skip gradient, weight / bias / batch-norm / PReLU gradients against float64 autograd."""
    _conv_block_grads(ops, pm, transposed, cin, cout, F, T, B, skip_c, bn, 2e-5, GTOL, causal=False)


@pytest.mark.parametrize("transposed,cx,cout,cin_total,ci_off,F,T,B", [
    (False, 32, 128, 32, 0, 33, 40, 3),       # conv, one S tile
    (False, 96, 160, 96, 0, 17, 29, 2),       # conv, ragged S and L tiles
    (True, 128, 48, 224, 96, 9, 35, 3),       # transposed conv, the skip half of the weight (ci_off > 0)
    (True, 160, 32, 160, 0, 5, 130, 2),       # transposed conv, ragged S tile
])
def test_wgrad_gauss_matches_four_product_kernel(ops, transposed, cx, cout, cin_total, ci_off, F, T, B):
    """The three-product (Gauss) weight gradient (idv_cconv2d_bwd_weight_gauss) against the four-product contraction
    (idv_cconv2d_bwd_weight), itself checked against float64 autograd in test_conv_block_grads: same fp32 operands, different
    association -> 2e-5 relative per tensor."""
    L = ops.L
    g = torch.Generator(device="cuda").manual_seed(11)
    Fo = 2 * F - 1 if transposed else (F - 1) // 2 + 1
    x = ops.Planar.empty(cx, F, B, T, T + 1, "cuda")
    dy = ops.Planar.empty(cout, Fo, B, T, T + 1, "cuda")
    for t in (x, dy):
        t.buf.normal_(generator=g)
        t.planes()[..., 0] = 0.0                                   # guard columns
    shape = (cin_total, cout, 5, 2) if transposed else (cout, cin_total, 5, 2)
    cs, cl = (cx, cout) if transposed else (cout, cx)
    assert L.lib().idv_cconv_wgrad_gauss_supported(L.i(cs), L.i(cl))
    outs = []
    for name, sizer, sargs in (("idv_cconv2d_bwd_weight", "idv_cconv_wgrad_work_floats", (L.i(cs), L.i(cl), L.i(B), L.i(x.Tp))),
                               ("idv_cconv2d_bwd_weight_gauss", "idv_cconv_wgrad_gauss_work_floats",
                                (L.i(cx), L.i(cout), L.i(int(transposed)), L.i(F), L.i(B), L.i(x.Tp), L.i(x.Jp), L.i(dy.Jp)))):
        n = int(ops._ll_fn(sizer)(*sargs))
        work = torch.empty(n, device="cuda")
        dwr = torch.full(shape, 7.0, device="cuda")              # rows outside [ci_off, ci_off + cx) must stay untouched
        dwi = torch.full(shape, 7.0, device="cuda")
        L.call(name, x.ptr(), L.i(cx), L.i(ci_off), dy.ptr(), L.i(cout), L.i(cin_total), L.i(int(transposed)), L.i(-1), L.i(F), L.i(B),
               L.i(x.Tp), L.i(x.Jp), L.i(dy.Jp), L.p(work), L.ll(n), L.p(dwr), L.p(dwi), L.stream_ptr())
        torch.cuda.synchronize()
        outs.append((dwr.cpu().double(), dwi.cpu().double()))
    for a, b in zip(outs[0], outs[1]):
        assert float((a - b).norm() / a.norm()) < 2e-5
    sl = outs[1][0][ci_off:ci_off + cx] if transposed else outs[1][0][:, ci_off:ci_off + cx]
    assert float((sl - 7.0).abs().min()) > 0 and float((outs[1][0] == 7.0).sum()) == outs[1][0].numel() - sl.numel()


def _conv_block_grads(ops, pm, transposed, cin, cout, F, T, B, skip_c, bn, ftol, gtol, causal=True):
    g = torch.Generator().manual_seed(3)
    dev = "cuda"
    cin_tot = cin + skip_c
    Fout = 2 * F - 1 if transposed else (F - 1) // 2 + 1
    Tout = T if causal else (T + 1 if transposed else T - 1)
    if transposed:
        blk = pm.Decoder(cin_tot, cout, (5, 2), (2, 1), (cout, Fout, T), (2, 0), causal=causal, if_bn=bn)
        conv = blk.transconv
    else:
        blk = pm.Encoder(cin_tot, cout, (5, 2), (2, 1), (cout, Fout, T), (2, 1) if causal else (2, 0), causal=causal)
        conv = blk.conv
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(rnd(g, *p_.shape, scale=0.3))
        blk.prelu.weight.fill_(0.2)
    blk = blk.to(dev)
    x = rnd(g, B, cin, F, T, 2)
    sk = rnd(g, B, skip_c, F, T, 2) if skip_c else None
    R = rnd(g, B, cout, Fout, Tout, 2)
    xp = ops.Planar.from_tensor5(x.to(dev), T + 2)
    xp.buf.requires_grad_(True)
    skp = None
    if sk is not None:
        skp = ops.Planar.from_tensor5(sk.to(dev), T + 2)
        skp.buf.requires_grad_(True)
    with torch.enable_grad():
        if transposed:
            z = blk.forward_planar(xp, True, skip=skp)
        elif bn:
            z = blk.forward_planar(xp, True)
        else:
            z = importlib.import_module("i-dccrn-vae_amd.autograd").conv_block(conv, None, None, xp, None, False)
        assert z.buf.grad_fn is not None
        loss = (z.tensor5() * R.to(dev)).sum()
        loss.backward()
    # oracle in float64
    sd = {k: leaf64(v.cpu()) for k, v in blk.state_dict().items() if v.dtype.is_floating_point and "running" not in k and k[-3:] not in ("Vrr", "Vri", "Vii")}
    x64 = leaf64(x)
    s64 = leaf64(sk) if sk is not None else None
    xin = x64 if s64 is None else torch.cat([x64, s64], 1)
    n = "transconv.tconv" if transposed else "conv.conv"
    args = (sd[f"{n}_re.weight"], sd[f"{n}_re.bias"], sd[f"{n}_im.weight"], sd[f"{n}_im.bias"])
    if transposed:
        y = O.complex_conv_transpose2d(xin, *args, (2, 1), (2, 0), causal)
    else:
        y = O.complex_conv2d(xin, *args, (2, 1), (2, 1) if causal else (2, 0), causal)
    if bn:
        st = O.cbn_batch_stats(y)
        y = O.cbn_whiten_affine(y, *st, sd["bn.gamma_rr"], sd["bn.gamma_ri"], sd["bn.gamma_ii"], sd["bn.beta_r"], sd["bn.beta_i"])
        y = O.prelu(y, sd["prelu.weight"])
    check("forward", z.tensor5(), y, ftol)
    (y * R.double()).sum().backward()
    check("dx", ops.rewrap(xp.buf.grad, xp).tensor5(), x64.grad, gtol)
    if s64 is not None:
        check("dskip", ops.rewrap(skp.buf.grad, skp).tensor5(), s64.grad, gtol)
    gx = ops.rewrap(xp.buf.grad, xp).planes()
    assert float(gx[..., 0].abs().max()) == 0.0                         # guard column of the gradient stays zero
    params = dict(blk.named_parameters())
    for k, v in sd.items():
        if k not in params:
            continue
        if not bn and (k.startswith("bn.") or k.startswith("prelu")):
            continue
        got = params[k].grad
        assert got is not None, k
        atol = 1e-4 * float(R.norm()) if (bn and k.endswith(".bias")) else 0.0     # exactly zero in exact arithmetic
        check(k, got, v.grad, gtol, atol)


# ----------------------------------------------------------------------------- complex LSTM (BPTT)
@pytest.mark.parametrize("H,I,T,B", [(16, 20, 7, 3), (128, 64, 12, 5), (128, 160, 30, 18), (48, 32, 6, 2), (96, 160, 9, 17),
                                     (384, 64, 5, 3), (768, 64, 9, 20), (384, 128, 12, 33),
                                     (128, 64, 1, 3), (128, 64, 2, 17), (128, 64, 5, 100)])     # one / two frames, 7 tiles
def test_clstm_grads(ops, cp, H, I, T, B):
    _clstm_grads(ops, cp, H, I, T, B, 2e-5, GTOL)


@pytest.mark.parametrize("H,I,T,B", [(384, 64, 9, 3), (768, 128, 7, 20), (384, 1280, 33, 32), (128, 1280, 25, 19)])
def test_clstm_grads_bf16x3(ops, cp, H, I, T, B):
    """bf16x3 training mode: the forward recurrence is a split-bf16 kernel (H = 384 / 768: persistent cooperative, 1, 2 and 4
    row tiles per workgroup; H = 128: register-resident), which also leaves the activated gates and cell states for the fp32
    BPTT kernels."""
    keep = ops.PRECISION
    ops.set_precision("bf16x3")
    try:
        _clstm_grads(ops, cp, H, I, T, B, 1e-4, 1e-3)
    finally:
        ops.set_precision(keep)


def _clstm_grads(ops, cp, H, I, T, B, ftol, gtol):
    g = torch.Generator().manual_seed(11)
    dev = "cuda"
    m = cp.ComplexLSTM(I, H, dev, num_layers=2)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(rnd(g, *p_.shape, scale=1.0 / H ** 0.5))
    m = m.to(dev)
    x = rnd(g, T, B, I, 2)
    R = rnd(g, T, B, H, 2)
    xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).to(dev))
    xp.buf.requires_grad_(True)
    with torch.enable_grad():
        out = m.forward_planar(xp)
        y = out.channel_slice(0, H).permute(1, 0, 2, 3)                 # [T, B, H, 2]
        (y * R.to(dev)).sum().backward()
    sd = {k: leaf64(v.cpu()) for k, v in m.state_dict().items()}
    x64 = leaf64(x)
    want = O.complex_lstm(x64, sd, "", 2)
    check("forward", y, want, ftol)
    (want * R.double()).sum().backward()
    gx = ops.rewrap(xp.buf.grad, xp).tensor5()                          # [B, I, 1, T, 2]
    check("dx", gx[:, :, 0].permute(2, 0, 1, 3), x64.grad, gtol)
    for k, p_ in m.named_parameters():
        check(k, p_.grad, sd[k].grad, gtol)


def test_cooperative_bptt_times_out_instead_of_hanging(ops, cp):
    """Safety net of the cooperative BPTT (lstm_bptt_stack2_f32.hip: both H = 128 layers in one launch): with one layer-1
    workgroup withheld (IDV_COOP_FAULT=1) every spin runs into its bound, both layers drain, the gate gradients are poisoned
    with NaN -- the backward returns within seconds, the status word says -3 -- and the next forward + backward is correct."""
    import os
    import time
    H, I, T, B = 128, 64, 12, 5
    dev = "cuda"
    g = torch.Generator().manual_seed(13)
    m = cp.ComplexLSTM(I, H, dev, num_layers=2)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(rnd(g, *p_.shape, scale=1.0 / H ** 0.5))
    m = m.to(dev)
    x = rnd(g, T, B, I, 2)
    lib = ops.L.lib()
    assert lib.idv_lstm_bptt_stack2_supported(H, B)

    def run():
        for p_ in m.parameters():
            p_.grad = None
        xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).to(dev))
        xp.buf.requires_grad_(True)
        with torch.enable_grad():
            m.forward_planar(xp).buf.sum().backward()
        torch.cuda.synchronize()
        # (the valid region: slack and row padding of the gradient buffer are never written)
        return ops.rewrap(xp.buf.grad, xp).tensor5().clone(), [p_.grad.clone() for p_ in m.parameters()]

    good = run()
    try:
        with torch.enable_grad():
            xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).to(dev))
            xp.buf.requires_grad_(True)
            loss = m.forward_planar(xp).buf.sum()
            torch.cuda.synchronize()
            os.environ["IDV_COOP_FAULT"] = "1"
            t0 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    finally:
        os.environ.pop("IDV_COOP_FAULT", None)
        # collected (and cleared) here whatever happens above: a stale sticky status would make the NEXT cooperative launch of
        # the process -- an unrelated test -- return -3
        torch.cuda.synchronize()
        status = lib.idv_coop_last_status(1)
    assert dt < 5.0, dt
    assert status == -3                                         # reported through the C ABI (and cleared)
    assert torch.isnan(xp.buf.grad).any()
    again = run()
    assert lib.idv_coop_last_status(1) == 0
    assert torch.isfinite(again[0]).all() and torch.equal(again[0], good[0])
    for a_, b_ in zip(again[1], good[1]):
        assert torch.equal(a_, b_)


# ----------------------------------------------------------------------------- dense, mask, STFT / ISTFT
def test_cdense_grads(ops, cp):
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    m = cp.ComplexDense(16, 40).to(dev)
    B, T = 3, 11
    x = rnd(g, B, T, 16, 2)
    R = rnd(g, B, 8, 5, T, 2)
    xp = ops.Planar.from_tensor5(x.permute(0, 2, 1, 3).unsqueeze(2).to(dev))
    xp.buf.requires_grad_(True)
    with torch.enable_grad():
        out = m.forward_planar(xp, 8, 5)
        (out.tensor5() * R.to(dev)).sum().backward()
    sd = {k: leaf64(v.cpu()) for k, v in m.state_dict().items()}
    x64 = leaf64(x)
    y = O.complex_dense(x64.reshape(B * T, 16, 2), sd["linear_read.weight"], sd["linear_read.bias"], sd["linear_imag.weight"],
                        sd["linear_imag.bias"]).reshape(B, T, 8, 5, 2).permute(0, 2, 3, 1, 4)
    check("forward", out.tensor5(), y, 2e-5)
    (y * R.double()).sum().backward()
    check("dx", ops.rewrap(xp.buf.grad, xp).tensor5()[:, :, 0].permute(0, 2, 1, 3), x64.grad)
    for k, p_ in m.named_parameters():
        check(k, p_.grad, sd[k].grad)


def test_mask_istft_stft_grads(ops, pm):
    """signal -> STFT -> (mask x STFT) -> ISTFT, gradients w.r.t. the mask AND the signal (reflect padding adjoint)."""
    g = torch.Generator().manual_seed(6)
    dev = "cuda"
    B, L = 2, 3200
    T = 1 + L // HOP
    sig = rnd(g, B, L, scale=0.1)
    M = rnd(g, B, 257, T, 2)
    R = rnd(g, B, L, scale=1.0)
    Rc = rnd(g, B, 257, T, 2)
    stft, istft = pm.STFT(NFFT, HOP, WIN, dev), pm.ISTFT(NFFT, HOP, WIN, dev)
    s_gpu = sig.to(dev).requires_grad_(True)
    mp = ops.Planar.from_tensor5(M.unsqueeze(1).to(dev))
    mp.buf.requires_grad_(True)
    with torch.enable_grad():
        X = stft.planar(s_gpu)
        assert X.buf.grad_fn is not None
        pred, predict = pm._predict_outputs(None, mp, X, "mask")
        y = istft.planar(pred)
        y2 = istft.planar(X)
        loss = (y * R.to(dev)).sum() + (torch.view_as_real(predict) * Rc.to(dev)).sum() + (y2 * R.to(dev)).sum() * 0.5
        loss.backward()
    s64, M64 = leaf64(sig), leaf64(M)
    X64 = O.stft(s64, NFFT, HOP, WIN)
    P = O.apply_mask(M64, X64)
    yo = O.istft(P, NFFT, HOP, WIN)
    yo2 = O.istft(X64, NFFT, HOP, WIN)
    check("waveform", y, yo, 2e-5)
    ((yo * R.double()).sum() + (P * Rc.double()).sum() + (yo2 * R.double()).sum() * 0.5).backward()
    check("dmask", ops.rewrap(mp.buf.grad, mp).tensor5()[:, 0], M64.grad)
    check("dsignal", s_gpu.grad, s64.grad)


# ----------------------------------------------------------------------------- reparameterisation + losses
def test_reparam_and_loss_grads(ops, pm, losses):
    nl, plm, sl = losses
    g = torch.Generator().manual_seed(8)
    dev = "cuda"
    B, T, H, ns = 2, 9, 8, 3
    lat = torch.cat([rnd(g, B, T, H, 2), rnd(g, B, T, H, 2, scale=0.3), rnd(g, B, T, H, 2, scale=0.8)], 2)   # miu | ls | delta
    lat[0, :3, 2 * H:, :] *= 4.0                                         # force the |delta| >= sigma guard branch
    lat2 = torch.cat([rnd(g, B, T, H, 2), rnd(g, B, T, H, 2, scale=0.3), rnd(g, B, T, H, 2, scale=0.8)], 2)
    eps_r, eps_i = rnd(g, B, ns, T, H), rnd(g, B, ns, T, H)
    Rz = rnd(g, B * ns, T, H, 2)
    enc = pm.pvae_dccrn_encoder_skip_prepare(O.net_params(True, 4), True, dev, H, NFFT, HOP, WIN, ns)
    lp = ops.Planar.from_tensor5(lat.permute(0, 2, 1, 3).unsqueeze(2).to(dev))
    lp2 = ops.Planar.from_tensor5(lat2.permute(0, 2, 1, 3).unsqueeze(2).to(dev))
    lp.buf.requires_grad_(True)

    def views(pl):
        out = []
        for k in range(3):
            v = pl.channel_slice(k * H, (k + 1) * H)
            v._idv, v._idv_off = pl, k * H
            out.append(v)
        return out
    with torch.enable_grad():
        z = enc._sample(lp, 0, (eps_r.to(dev), eps_i.to(dev)))
        miu, ls, dl = views(lp)
        miu2, ls2, dl2 = views(lp2)
        nll = nl.standard_nsvae_loss_true_kl(1.0, 0, 1.0, 0.5, H, ns, 1, 'original', 'False', [], 'both')
        kl = nll.cal_kl(miu, miu2, ls, ls2, dl, dl2, None)
        pl_ = plm.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', [1, 1, 0], ns)
        klp = pl_.cal_kl_arbi_prior(miu, miu2, ls, ls2, dl, dl2)
        dis, _, _ = nll.miu_dis_loss(miu2, miu2, miu, miu)
        loss = (z * Rz.to(dev)).sum() + 0.7 * kl + 0.3 * klp + 0.9 * dis
        loss.backward()
    L64 = leaf64(lat)
    m64, s64, d64 = L64[:, :, :H], L64[:, :, H:2 * H], L64[:, :, 2 * H:]
    l2 = lat2.double()
    zo = O.reparameterization(m64, s64, d64, ns, eps_r.double(), eps_i.double())
    check("z", z, zo, 2e-5)
    klo = O.complex_kl(m64, l2[:, :, :H], s64, l2[:, :, H:2 * H], d64, l2[:, :, 2 * H:], 1e-10).mean()
    klpo = O.complex_kl(m64, l2[:, :, :H], s64, l2[:, :, H:2 * H], d64, l2[:, :, 2 * H:], 1e-9).mean()
    diso = 2 * torch.sqrt(((l2[:, :, :H] - m64) ** 2).mean(dim=(0, 1)).sum())
    assert abs(float(kl) - float(klo)) < 1e-4 * abs(float(klo)) and abs(float(dis) - float(diso)) < 1e-4 * float(diso)
    ((zo * Rz.double()).sum() + 0.7 * klo + 0.3 * klpo + 0.9 * diso).backward()
    got = ops.rewrap(lp.buf.grad, lp).tensor5()[:, :, 0].permute(0, 2, 1, 3)          # [B, T, 3H, 2]
    check("dlat", got, L64.grad)

    # SI-SNR and the two STFT reconstruction terms
    src, est = rnd(g, 4, 1600, scale=0.1), rnd(g, 4, 1600, scale=0.1)
    est = src + 0.3 * est
    P, Or = rnd(g, 4, 257, 9, 2), rnd(g, 2, 257, 9, 2)
    e_gpu = est.to(dev).requires_grad_(True)
    p_gpu = P.to(dev).requires_grad_(True)
    with torch.enable_grad():
        L_ = nl.ete_train_se_loss([0.3, 0.5, 1.0])
        tot = L_.final_ete_loss(torch.view_as_complex(p_gpu), Or.to(dev), src.to(dev), e_gpu)
        tot[0].backward()
    e64, p64 = leaf64(est), leaf64(P)
    want = O.multiple_recon_loss(p64, Or.double().repeat_interleave(2, 0), src.double(), e64, [0.3, 0.5, 1.0])
    assert abs(float(tot[0]) - float(want[0])) < 1e-4 * abs(float(want[0]))
    want[0].backward()
    check("d est", e_gpu.grad, e64.grad)
    check("d pred", p_gpu.grad, p64.grad)


# ----------------------------------------------------------------------------- reference gradient fixtures
def load_synth(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(O.synth_state_dict(shapes, seed), strict=True)
    return module.cuda()


def summarize(t, limit=16384, cap=8192):
    t = t.detach().reshape(-1)
    if t.numel() <= limit:
        return t
    return t[::-(-t.numel() // cap)]


def check_grads(d, prefix, module, tol=1e-3, tol1=None, errs=None):
    """Every parameter gradient of `module` against the reference's (summarised gradient + full L2 norm); tol1: tolerance
    for one-element tensors (default tol); errs: dict that receives the per-tensor error of every multi-element tensor."""
    n = 0
    worst = (0.0, "")
    lim = (int(d["sum_limit"]), int(d["sum_cap"])) if "sum_limit" in d.files else (16384, 8192)
    for k, p_ in module.named_parameters():
        key = f"g:{prefix}{k}"
        if key not in d.files:
            assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, f"{k}: reference has no gradient"
            continue
        assert p_.grad is not None, f"{k}: no gradient"
        want, wn = T_(d[key]).double(), float(d[f"n:{prefix}{k}"])
        got = summarize(p_.grad, *lim).cpu().double()
        scale = max(wn * (want.numel() / p_.numel()) ** 0.5, 1e-12)
        tk = tol1 if (tol1 is not None and p_.numel() == 1) else tol
        err = float((got - want).norm()) / scale
        gn = float(p_.grad.double().norm())
        sib = f"n:{prefix}{k[:-4]}weight"
        if k.endswith(".bias") and sib in d.files and wn < 1e-4 * float(d[sib]):
            # a bias in front of a batch norm has an exactly-zero true gradient (both sides hold rounding noise):
            # compare against the scale of the layer's weight gradient instead
            assert gn < 1e-3 * float(d[sib]), (k, gn, wn, float(d[sib]))
        else:
            assert err < tk, f"{k}: rel err {err:.2e}"
            assert abs(gn - wn) < tk * wn, f"{k}: norm {gn} vs {wn}"
            if p_.numel() > 1 or tol1 is None:
                worst = max(worst, (err, k))
                if errs is not None:
                    errs[k] = err
        n += 1
    assert n > 0
    return worst


def check_adam(d, prefix, module, lr=1e-3, wd=1e-3, tol=1e-2):
    """One Adam step reproduces the reference's updated weights.  The first Adam step is lr * g / (|g| + 1e-8): sign-like,
    so an element whose gradient is of the order of the 1e-8 epsilon (or of fp32 rounding noise) moves by a different
    fraction of lr on the two sides even when the gradients agree to 1e-4 -- hence 1e-2 on the update vector (the
    gradients themselves are held to 1e-3 by check_grads)."""
    params = [p_ for p_ in module.parameters() if p_.grad is not None]
    before = {k: p_.detach().clone() for k, p_ in module.named_parameters()}
    torch.optim.Adam(params, lr=lr, weight_decay=wd).step()
    seen = 0
    for k, p_ in module.named_parameters():
        key = f"a:{prefix}{k}"
        if key in d.files:
            want = T_(d[key]).double()
            got = summarize(p_).cpu().double()
            step = (want - summarize(before[k]).cpu().double())
            # the update itself (lr * sign-like step) must match, not just the weights
            assert float(((got - summarize(before[k]).cpu().double()) - step).norm()) < tol * float(step.norm()) + 1e-9, k
            seen += 1
    assert seen > 0


def test_grad_dccrn_reference(pm, losses, golden):
    """supervised_dccrn/train.py:233-243: model(noisy) -> final_ete_loss -> backward -> Adam.step, mini DCCRN-CL."""
    d = golden("grad_dccrn_mini")
    nl, _, _ = losses
    base, seed = int(d["base"]), int(d["seed"])
    np_ = O.net_params(True, base)
    m = load_synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None), seed)
    m.train()
    x = T_(d["x"]).cuda().requires_grad_(True)
    clean_ref = T_(d["clean_ref"]).cuda()
    w = [float(v) for v in d["weights"]]
    with torch.enable_grad():
        est, pred = m(x, train=True)
        loss = nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(clean_ref), clean_ref, est)
        loss[0].backward()
    assert relerr(est.detach().cpu(), T_(d["est"])) < 1e-4
    for a, b in zip(loss, T_(d["loss"])):
        assert abs(float(a) - float(b)) < 2e-4 * max(1.0, abs(float(b)))
    assert relerr(x.grad.cpu(), T_(d["gx"])) < 1e-3
    worst = check_grads(d, "", m)
    print("worst parameter-gradient error", worst)
    check_adam(d, "", m)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_grad_dccrn_reference_full_width(pm, losses, golden, ops, precision):
    """The same train step at the reference's FULL width (base 32: every kernel at the benchmark's tile counts) on 1 s
    utterances against gradients written by the REAL reference (make_golden.py gradfull: 1024-element subsample + full L2
    norm per tensor), in both arithmetic modes.  Tolerance 1e-2 (3e-2 for one-element tensors) for the reason given in
    test_full_width_train_step_grads: a handful of PReLU pre-activations within rounding of zero take the other branch
    (measured worst multi-element tensor: 4.4e-3).  bf16x3: the forward rounding is 1e-5 instead of 1e-6, ten times as many
    pre-activations lie within it, the deviation grows with the square root: 3e-2 (measured 8.8e-3)."""
    d = golden("grad_dccrn_full")
    nl, _, _ = losses
    base, seed = int(d["base"]), int(d["seed"])
    np_ = O.net_params(True, base)
    keep = ops.PRECISION
    ops.set_precision(precision)
    try:
        m = load_synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None), seed)
        m.train()
        x = T_(d["x"]).cuda().requires_grad_(True)
        clean_ref = T_(d["clean_ref"]).cuda()
        w = [float(v) for v in d["weights"]]
        with torch.enable_grad():
            est, pred = m(x, train=True)
            loss = nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(clean_ref), clean_ref, est)
            loss[0].backward()
    finally:
        ops.set_precision(keep)
    assert relerr(est.detach().cpu(), T_(d["est"])) < 1e-4
    for a, b in zip(loss, T_(d["loss"])):
        assert abs(float(a) - float(b)) < 2e-4 * max(1.0, abs(float(b)))
    tol = 1e-2 if precision == "fp32" else 3e-2
    assert relerr(x.grad.cpu(), T_(d["gx"])) < tol
    errs = {}
    worst = check_grads(d, "", m, tol=tol, tol1=3 * tol, errs=errs)
    print("worst parameter-gradient error vs the reference", worst)
    # A flipped PReLU element near the output moves EVERY upstream gradient by the same relative amount, so the distribution
    # is flat: measured fp32 median 2.1e-3 / 90th percentile 2.6e-3 / max 4.1e-3 with the three-product conv kernel, 2.6e-3 /
    # 3.1e-3 / 4.4e-3 with cgemm_kernel (IDV_GAUSS=0).  Caps = measured + 2x.
    e = sorted(errs.values())
    med, p90 = e[len(e) // 2], e[(9 * len(e)) // 10]
    print(f"{precision}: {len(e)} tensors, median {med:.2e}, 90th percentile {p90:.2e}, max {e[-1]:.2e}")
    _dump(f"grad_ref_full_{precision}", errs)
    med_cap, p90_cap = FULL_REF_CAPS[precision]
    assert med < med_cap and p90 < p90_cap, (med, p90)


FULL_REF_CAPS = {"fp32": (5e-3, 6e-3), "bf16x3": (1.5e-2, 2e-2)}


def _dump(name, obj):
    """Per-tensor error tables for the profiles / tolerance bookkeeping (only when run on the GPU box by gpurun)."""
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(obj, f, indent=0, sort_keys=True)


# The VAE train steps against the reference's own loss.backward(): "mini" = reduced width (every gradient element + one Adam
# step); "full" = the reference's FULL width (base 32, zdim 128: LSTM hidden 384 / 768 on the persistent recurrences, per-step or
# cooperative BPTT, the K = 1280 input projections with their data / weight gradients, the repeated-skip decoder at its real
# channel counts), 1 s utterances, B = 2 x ns = 2, make_golden.py gradvaefull: 1024-element subsample + full L2 norm per tensor.
VAE_GRAD_CASES = [("mini", "fp32"), ("full", "fp32"), ("full", "bf16x3")]
# full width: a handful of PReLU pre-activations within rounding of zero take the other branch and move every upstream gradient
# (test_full_width_train_step_grads); tolerances = those of the supervised full-width fixture, the measured values are printed
VAE_GRAD_TOL = {("mini", "fp32"): (1e-3, None, 2e-4), ("full", "fp32"): (1e-2, 3e-2, 5e-4), ("full", "bf16x3"): (3e-2, 9e-2, 2e-3)}


class _Precision:
    def __init__(self, ops, mode):
        self.ops, self.mode = ops, mode

    def __enter__(self):
        self.keep = self.ops.PRECISION
        self.ops.set_precision(self.mode)

    def __exit__(self, *a):
        self.ops.set_precision(self.keep)


def _vae_grad_report(name, tag, precision, errs):
    e = sorted(errs.values())
    if e:
        print(f"{name}[{tag},{precision}]: {len(e)} tensors, median {e[len(e) // 2]:.2e}, 90th percentile {e[(9 * len(e)) // 10]:.2e}, "
              f"max {e[-1]:.2e}")
        _dump(f"grad_ref_{name}_{tag}_{precision}", errs)


@pytest.mark.parametrize("tag,precision", VAE_GRAD_CASES)
def test_grad_cvae_reference(pm, losses, golden, ops, tag, precision):
    """pretrained_vaes/train.py:281-301: encoder -> reparameterised z -> decoder (zero skips) -> ELBO -> backward."""
    d = golden(f"grad_cvae_{tag}")
    tol, tol1, ltol = VAE_GRAD_TOL[(tag, precision)]
    _, plm, _ = losses
    base, seed, zdim, ns = int(d["base"]), int(d["seed"]), int(d["zdim"]), int(d["ns"])
    np_ = O.net_params(True, base)
    enc = load_synth(pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns), seed)
    dec = load_synth(pm.pvae_dccrn_decoder_skip_prepare(np_, True, "cuda", ns, zdim, NFFT, HOP, WIN, "real_imag", SKIP), seed + 1)
    x = T_(d["x"]).cuda()
    B, L = x.shape
    eps = (T_(d["eps_r"]).cuda(), T_(d["eps_i"]).cuda())
    w = [float(v) for v in d["weights"]]
    with torch.enable_grad(), _Precision(ops, precision):
        z, miu, ls, dl, skiper, C, F, stft_x = enc(x, train=True, eps=eps)
        recon, pred = dec(stft_x, z, skiper, C, F, train=True)
        xr = x.unsqueeze(1).repeat(1, ns, 1).view(B * ns, L)
        sx = stft_x.detach().unsqueeze(1).repeat(1, ns, 1, 1, 1).view(B * ns, stft_x.shape[1], stft_x.shape[2], 2)
        pl = plm.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', w, ns)
        lo = pl.cal_loss(xr, recon, sx, pred, miu, ls, dl, z, 5)
        lo[0].backward()
    got = torch.stack([torch.as_tensor(float(v)) for v in (lo[0], lo[1], lo[2], lo[4], lo[5], lo[6])])
    for a, b in zip(got, T_(d["loss"])):
        assert abs(float(a) - float(b)) < ltol * max(1.0, abs(float(b))), (float(a), float(b))
    errs = {}
    print("worst", check_grads(d, "enc.", enc, tol, tol1, errs), check_grads(d, "dec.", dec, tol, tol1, errs))
    _vae_grad_report("cvae", tag, precision, errs)
    if tag == "mini":
        check_adam(d, "enc.", enc)
        check_adam(d, "dec.", dec)


@pytest.mark.parametrize("tag,precision", VAE_GRAD_CASES)
def test_grad_nsvae_reference(pm, losses, golden, ops, tag, precision):
    """train_nsvae.py:487-574: frozen clean / noise encoders (eval, no_grad), trainable noisy encoder, nsvae KL loss."""
    d = golden(f"grad_nsvae_{tag}")
    tol, tol1, ltol = VAE_GRAD_TOL[(tag, precision)]
    nl, _, _ = losses
    base, seed, zdim, ns = int(d["base"]), int(d["seed"]), int(d["zdim"]), int(d["ns"])
    np_ = O.net_params(True, base)
    ce = load_synth(pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns), seed + 4)
    ne = load_synth(pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns), seed + 5)
    ye = load_synth(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns, 2), seed + 6)
    clean, noise = T_(d["clean"]).cuda(), T_(d["noise"]).cuda()
    noisy = clean + noise
    e = [T_(d[f"eps{i}"]).cuda() for i in range(8)]
    with _Precision(ops, precision):
        with torch.no_grad():
            zc, mc, lc, dc, skc, _, _, _ = ce(clean, train=False, eps=(e[0], e[1]))
            zn, mn, ln_, dn, skn, _, _, _ = ne(noise, train=False, eps=(e[2], e[3]))
        with torch.enable_grad():
            zs, ms, ls_, ds, znn, mnn, lnn, dnn, sky, C, F, stft_y = ye(noisy, train=True, eps=(e[4], e[5], e[6], e[7]))
            L_ = nl.standard_nsvae_loss_true_kl(1.0, 0, 1.0, 0.5, zdim, ns, 2, 'original', 'False', [], 'both')
            out = L_.final_nsvae_loss(mc, mn, ms, mnn, lc, ln_, ls_, lnn, dc, dn, ds, dnn, zs, znn, skc, skn, sky)
            out[0].backward()
    for a, b in zip(out[:6], T_(d["loss"])):
        assert abs(float(a) - float(b)) < 1.5 * ltol * max(1.0, abs(float(b))), (float(a), float(b))
    errs = {}
    print("worst", check_grads(d, "noisy.", ye, tol, tol1, errs))
    _vae_grad_report("nsvae", tag, precision, errs)
    if tag == "mini":
        check_adam(d, "noisy.", ye)
    assert all(p_.grad is None for p_ in ce.parameters())


@pytest.mark.parametrize("tag,precision", VAE_GRAD_CASES)
def test_grad_twophase_reference(pm, losses, golden, ops, tag, precision):
    """train_second_phase_decoder.py:376-433: frozen noisy encoder (eval), decoder with repeated real skips, SI-SNR."""
    d = golden(f"grad_twophase_{tag}")
    tol, tol1, ltol = VAE_GRAD_TOL[(tag, precision)]
    nl, _, _ = losses
    base, seed, zdim, ns = int(d["base"]), int(d["seed"]), int(d["zdim"]), int(d["ns"])
    np_ = O.net_params(True, base)
    ye = load_synth(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns, 2), seed + 6)
    de = load_synth(pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, "cuda", ns, zdim, NFFT, HOP, WIN, "mask", True, SKIP, False),
                    seed + 7)
    for p_ in ye.parameters():
        p_.requires_grad = False
    clean, noise = T_(d["clean"]).cuda(), T_(d["noise"]).cuda()
    noisy = clean + noise
    B, L = clean.shape
    e = [T_(d[f"eps{i}"]).cuda() for i in range(4)]
    with torch.enable_grad(), _Precision(ops, precision):
        r = ye(noisy, train=False, eps=tuple(e))
        zs, sky, C, F, stft_y = r[0], r[8], r[9], r[10], r[11]
        rec, prd = de(stft_y, zs, sky, C, F, train=True, pad='sig')
        sxc = ye.stft(clean).unsqueeze(1).repeat(1, ns, 1, 1, 1).view(B * ns, stft_y.shape[1], stft_y.shape[2], 2)
        cb = clean.unsqueeze(1).repeat(1, ns, 1).view(B * ns, L)
        tl = nl.two_phase_loss([0, 0, 1], 1.0, zdim, 1)
        l2 = tl.phase_2_loss(prd, sxc, cb, rec, None, None, None, None)
        l2[0].backward()
    rerr = relerr(rec.detach().cpu(), T_(d["recon"]))
    print(f"twophase[{tag},{precision}] waveform rel err {rerr:.2e}")
    assert rerr < (1e-4 if precision == "fp32" else 1e-3)
    for a, b in zip(l2[:4], T_(d["loss"])):
        assert abs(float(a) - float(b)) < ltol * max(1.0, abs(float(b))), (float(a), float(b))
    errs = {}
    print("worst", check_grads(d, "dec.", de, tol, tol1, errs))
    _vae_grad_report("twophase", tag, precision, errs)
    if tag == "mini":
        check_adam(d, "dec.", de)


def test_eval_after_train_uses_updated_running_stats(pm, ops):
    """eval -> train-mode forward(s) -> eval must see the running statistics the train step wrote
    (they are updated through raw pointers; the fold / pack caches are keyed on a generation counter)."""
    np_ = O.net_params(True, 4)
    m = load_synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None), 9)
    g = torch.Generator().manual_seed(1)
    x1, x2 = rnd(g, 2, 1600, scale=0.1), rnd(g, 2, 1600, scale=0.3) + 0.05
    with torch.no_grad():
        a, _ = m(x1.cuda(), train=False)
        m(x2.cuda(), train=True)
        b, _ = m(x1.cuda(), train=False)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    want, _, _ = O.dccrn_forward(x1, sd, np_, True, NFFT, HOP, WIN, SKIP, "mask", False)
    assert relerr(b.cpu(), want) < 1e-4
    assert relerr(a.cpu(), want) > 1e-3               # the statistics really changed
    with torch.enable_grad():                         # and through the autograd path
        m(x2.cuda() * 0.5, train=True)
    with torch.no_grad():
        c, _ = m(x1.cuda(), train=False)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    want2, _, _ = O.dccrn_forward(x1, sd, np_, True, NFFT, HOP, WIN, SKIP, "mask", False)
    assert relerr(c.cpu(), want2) < 1e-4


def test_end_to_end_encoder_decoder_grads_with_repeated_skips(pm, losses):
    """Trainable NSVAE encoder -> reparameterised z -> decoder with real skips repeated num_samples times (pad='sig', mask)
    -> SI-SNR: gradients reach the ENCODER through z and through the repeated skip connections (RepeatBatchFn sums the
    copies).  Against torch.autograd through the oracle in float64."""
    nl, _, _ = losses
    base, zdim, ns, B, L = 4, 32, 2, 2, 1600
    T = 1 + L // HOP
    np_ = O.net_params(True, base)
    enc = load_synth(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns, 2), 31)
    dec = load_synth(pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, "cuda", ns, zdim, NFFT, HOP, WIN, "mask", True, SKIP, False), 32)
    g = torch.Generator().manual_seed(2)
    x = rnd(g, B, L, scale=0.1)
    clean = x + rnd(g, B, L, scale=0.03)
    eps = [rnd(g, B, ns, T, zdim) for _ in range(4)]
    with torch.enable_grad():
        r = enc(x.cuda(), train=True, eps=tuple(e.cuda() for e in eps))
        rec, prd = dec(r[11], r[0], r[8], r[9], r[10], train=True, pad='sig')
        loss = nl.two_phase_loss([0, 0, 1], 1.0, zdim, 1).phase_2_loss(prd, r[11].detach(), clean.cuda(), rec, None, None, None, None)[0]
        loss.backward()
    sd_e = {k: leaf64(v.cpu()) for k, v in enc.state_dict().items() if v.dtype.is_floating_point}
    sd_d = {k: leaf64(v.cpu()) for k, v in dec.state_dict().items() if v.dtype.is_floating_point}
    oe = O.vae_encoder_forward(x.double(), sd_e, np_, True, zdim, NFFT, HOP, WIN, ns, 2, [e.double() for e in eps], True)
    o_rec, o_pred = O.vae_decoder_forward(oe["stft_x"], oe["z_speech"], oe["skiper"], 8 * base, 5, sd_d, np_, True, ns, NFFT, HOP, WIN,
                                          "mask", SKIP, "sig", True, True)
    check("recon", rec, o_rec, 1e-4)
    ol = O.si_snr(clean.double().repeat_interleave(ns, 0), o_rec)
    assert abs(float(loss.detach()) - float(ol)) < 1e-4 * abs(float(ol))
    ol.backward()
    n = 0
    for mod, sd in ((enc, sd_e), (dec, sd_d)):
        for k, p_ in mod.named_parameters():
            want = sd[k].grad
            if want is None or p_.grad is None:
                assert (want is None or float(want.abs().max()) == 0.0) and (p_.grad is None or float(p_.grad.abs().max()) == 0.0), k
                continue
            if k.endswith("conv_re.bias") or k.endswith("conv_im.bias"):
                continue
            check(k, p_.grad, want, 1e-3)
            n += 1
    assert n > 60


def test_grad_dccrn_datanorm_reference(pm, losses, golden):
    """Training WITH the reference's --data_norm (model/pvae_module.py:217-221, :235-238; VERDICT r2 'missing' 5b): the mini
    DCCRN-CL train step against loss and parameter gradients written by the REAL reference (make_golden.py datanorm)."""
    d = golden("dccrn_datanorm_mini")
    nl, _, _ = losses
    np_ = O.net_params(True, 4)
    mean, std = T_(d["data_mean"]), T_(d["data_std"])
    m = pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, mean, std)
    sd = O.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if k not in ("data_mean", "data_std")}, int(d["seed"]))
    sd["data_mean"], sd["data_std"] = mean, std
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    m.train()
    x, clean_ref = T_(d["x"]).cuda(), T_(d["train_clean_ref"]).cuda()
    w = [float(v) for v in d["train_weights"]]
    with torch.enable_grad():
        est, pred = m(x, train=True)
        loss = nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(clean_ref), clean_ref, est)
        loss[0].backward()
    assert relerr(est.detach().cpu(), T_(d["train_est"])) < 1e-4
    for a, b in zip(loss, T_(d["train_loss"])):
        assert abs(float(a) - float(b)) < 2e-4 * max(1.0, abs(float(b)))
    print("worst parameter-gradient error", check_grads(d, "train:", m))


def test_noncausal_dccrn_train_step_grads(pm, losses):
    """The non-causal DCCRN (model/net_config.py; VERDICT r2 'missing' 5a) as a TRAIN step: forward(train=True) + final_ete_loss +
    backward through the HIP autograd path against torch.autograd through the oracle in float64 -- loss, input gradient and every
    parameter gradient (mini width: no PReLU pre-activation sits within rounding of zero here, so 1e-3 holds)."""
    nl, _, _ = losses
    np_ = O.net_params(False, 4)
    m = load_synth(pm.DCCRN_(NFFT, HOP, np_, False, "cuda", WIN, SKIP, "mask", False, None, None), 31)
    m.train()
    g = torch.Generator().manual_seed(5)
    x = rnd(g, 2, 2400, scale=0.1)
    c = x + rnd(g, 2, 2400, scale=0.05)
    xg = x.cuda().requires_grad_(True)
    w = [0.3, 0.2, 1.0]
    with torch.enable_grad():
        est, pred = m(xg, train=True)
        loss = nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(c.cuda()), c.cuda(), est)[0]
        loss.backward()
    sd = {k: v.detach().cpu().double().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    for k in list(sd):
        if ".bn.running" in k or k.endswith((".Vrr", ".Vri", ".Vii")):
            sd[k] = sd[k].detach()
    x64 = x.double().clone().requires_grad_(True)
    o_est, o_pred, _ = O.dccrn_forward(x64, sd, np_, False, NFFT, HOP, WIN, SKIP, "mask", True, O.BNState())
    o_loss = O.multiple_recon_loss(o_pred, O.stft(c.double(), NFFT, HOP, WIN), c.double(), o_est, w)[0]
    o_loss.backward()
    assert tuple(est.shape) == tuple(o_est.shape)
    assert abs(float(loss.detach()) - float(o_loss)) < 2e-4 * abs(float(o_loss))
    check("waveform", est, o_est.detach(), 1e-4)
    check("input gradient", xg.grad, x64.grad, 1e-3)
    n = 0
    for k, p_ in m.named_parameters():
        want = sd[k].grad
        if want is None:
            assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, k
            continue
        if k.endswith("conv_re.bias") or k.endswith("conv_im.bias"):
            continue                                    # true-zero gradients in front of a batch norm
        check(k, p_.grad, want, 1e-3 if p_.numel() > 1 else 3e-3)
        n += 1
    assert n > 100


_FULL_WIDTH_ORACLE = {}


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_full_width_train_step_grads(pm, losses, ops, precision, golden):
    """The DCCRN-CL train step at the reference's FULL width (base 32: up to 512-channel blocks, every kernel at the tile
    counts the benchmark runs) on 1 s utterances: loss, input gradient and every parameter gradient against
    torch.autograd through the oracle in float64 on the CPU.

    Tolerance 1e-2, and why (tests/tools/err_probe_*.py print all of this on a GPU box): every block's gradients agree with
    float64 to 1e-7 .. 1e-6 on random data (err_probe.py, err_probe_bn.py, err_probe_lstm.py) -- but PReLU is not smooth:
    of the ~10^7 pre-activations of a block a handful lie within the forward rounding error (1e-6) of zero, take the other
    branch than in float64, and change that block's data gradient by sqrt(flips / N) ~ 1e-4 .. 1e-3 in L2
    (err_probe_block.py: blocks in isolation on the real activations are at 1e-6 except where an element flipped, e.g.
    decoders.4 2e-4, encoders.3 1e-3; the float32 oracle shows the same jumps at other blocks, encoders.2 / .5).  The
    float32 oracle's own deviation from float64 is printed beside ours (4e-4 .. 1.5e-3).  Against the reference's fp32
    gradients the mini fixtures hold 1e-3 (worst 4e-4).

    bf16x3: the same step in split-bf16 training mode (conv forward + data gradient on the bf16 MFMA kernels) under the
    same bars; the oracle runs once for both."""
    nl, _, _ = losses
    keep = ops.PRECISION
    ops.set_precision(precision)
    try:
        _full_width(pm, nl, golden("grad_dccrn_full"))
    finally:
        ops.set_precision(keep)


def _full_width(pm, nl, ref):
    # the step of the reference fixture (tests/golden/grad_dccrn_full.npz): the gradients the REFERENCE's own float32 arithmetic
    # produced for it are the yardstick below
    np_ = O.net_params(True, int(ref["base"]))
    m = load_synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None), int(ref["seed"]))
    m.train()
    x = T_(ref["x"]).float()
    c = T_(ref["clean_ref"]).float()
    xg = x.cuda().requires_grad_(True)
    w = [float(v) for v in ref["weights"]]
    with torch.enable_grad():
        est, pred = m(xg, train=True)
        loss = nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(c.cuda()), c.cuda(), est)[0]
        loss.backward()
    if not _FULL_WIDTH_ORACLE:
        sd = {k: v.detach().cpu().double().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
        for k in list(sd):
            if ".bn.running" in k or k.endswith((".Vrr", ".Vri", ".Vii")):
                sd[k] = sd[k].detach()
        # the running buffers were overwritten by the train step; the train-mode oracle does not read them
        x64 = x.double().clone().requires_grad_(True)
        o_est, o_pred, _ = O.dccrn_forward(x64, sd, np_, True, NFFT, HOP, WIN, SKIP, "mask", True, O.BNState())
        o_loss = O.multiple_recon_loss(o_pred, O.stft(c.double(), NFFT, HOP, WIN), c.double(), o_est, w)[0]
        o_loss.backward()
        # yardstick: the SAME oracle in float32 (what the reference's fp32 torch arithmetic amounts to) against float64
        sd32 = {k: v.detach().float().clone().requires_grad_(v.requires_grad) for k, v in sd.items()}
        x32 = x.clone().requires_grad_(True)
        e32, p32, _ = O.dccrn_forward(x32, sd32, np_, True, NFFT, HOP, WIN, SKIP, "mask", True, O.BNState())
        O.multiple_recon_loss(p32, O.stft(c, NFFT, HOP, WIN), c, e32, w)[0].backward()
        _FULL_WIDTH_ORACLE.update(sd=sd, x64=x64, o_est=o_est.detach(), o_loss=float(o_loss), sd32=sd32, x32=x32)
    sd, x64, o_est, o_loss, sd32, x32 = (_FULL_WIDTH_ORACLE[k] for k in ("sd", "x64", "o_est", "o_loss", "sd32", "x32"))
    assert abs(float(loss.detach()) - o_loss) < 2e-4 * abs(o_loss)
    check("waveform", est, o_est, 1e-4)

    def rel(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).norm() / (b.norm() + 1e-30))
    ours, ref32 = rel(xg.grad, x64.grad), rel(x32.grad, x64.grad)
    print(f"input gradient vs float64: HIP {ours:.2e}, float32 oracle {ref32:.2e}")
    assert ours < 1e-2
    n, worst = 0, (0.0, "", 0.0)
    table, sq, cand = {}, [0.0, 0.0, 0.0, 0.0], []
    lim = (int(ref["sum_limit"]), int(ref["sum_cap"])) if "sum_limit" in ref.files else (16384, 8192)
    for k, p_ in m.named_parameters():
        want = sd[k].grad
        if want is None:
            assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, k
            continue
        if k.endswith("conv_re.bias") or k.endswith("conv_im.bias"):
            continue                                    # true-zero gradients in front of a batch norm
        ours, ref32 = rel(p_.grad, want), rel(sd32[k].grad, want)
        # a one-element gradient (the shared PReLU slope: a 10^7-term sum with heavy cancellation) is a single noise
        # sample on either side, not an average over elements: 3e-2
        tol = 3e-2 if p_.numel() == 1 else 1e-2
        assert ours < tol, (k, ours, ref32)
        # ... and the bar that is ASSERTED, not argued (from an error model, not from the last
        # measurement).  The yardstick is what the REFERENCE's own float32 arithmetic delivers on this very step: the gradients it wrote
        # (fixture: a strided subsample + the full norm per tensor) against float64 -- `torch` -- beside the float32 oracle's deviation
        # `oracle32`.  PReLU-flip noise is an independent sample in each float32 evaluation, so per tensor the HIP path may deviate at
        # most YARD[..][0] times as far as the WORSE of the two float32 yardsticks (floor: YARD[..][1], for tensors where both happen to
        # be flip-free: that multiple of the reference's whole-vector deviation); the whole gradient vector, where the flip noise
        # averages out, at most YARD[..][2] times.
        if p_.numel() > 1:
            fx, wn_ref = T_(ref[f"g:{k}"]).double(), float(ref[f"n:{k}"])
            w_sub = summarize(want, *lim).cpu().double()
            scale = max(wn_ref * (fx.numel() / p_.numel()) ** 0.5, 1e-30)
            torch_dev = float((fx - w_sub).norm()) / scale
            ours_sub = float((summarize(p_.grad, *lim).cpu().double() - w_sub).norm()) / scale
            cand.append((k, max(ours, ours_sub), max(torch_dev, ref32)))
            wn = float(want.double().norm())
            sq[0] += (ours * wn) ** 2
            sq[1] += (ref32 * wn) ** 2
            sq[2] += wn ** 2
            sq[3] += (torch_dev * wn) ** 2
            table[k] = (ours, ref32, torch_dev, ours_sub)
        else:
            table[k] = (ours, ref32)
        worst = max(worst, (ours, k, ref32))
        n += 1
    print("worst parameter gradient (HIP vs float64, name, float32 oracle vs float64)", worst)
    assert n > 100
    # all multi-element parameter gradients as ONE vector: the flip noise averages out, a kernel error would not
    tot_ours, tot_ref, tot_torch = (sq[0] / sq[2]) ** 0.5, (sq[1] / sq[2]) ** 0.5, (sq[3] / sq[2]) ** 0.5
    print(f"whole gradient vector vs float64: HIP {tot_ours:.2e}, float32 oracle {tot_ref:.2e}, the reference's float32 torch {tot_torch:.2e}")
    _dump(f"grad_f64_full_{ops_precision()}", {"table": table, "total": [tot_ours, tot_ref, tot_torch]})
    k_, floor_mult, k_tot = YARD[ops_precision()]
    viol = [(k, o, y) for k, o, y in cand if o > max(k_ * y, floor_mult * tot_torch)]
    assert not viol, (viol, tot_torch)
    assert tot_ours <= k_tot * max(tot_ref, tot_torch), (tot_ours, tot_ref, tot_torch)


# (per-tensor factor, per-tensor floor as a multiple of the REFERENCE's whole-vector deviation, whole-vector factor).
# The yardstick is what the reference's own float32 arithmetic delivers on this step (an error model, not the last
# measurement): its gradients (fixture grad_dccrn_full.npz) are 1.16e-3 from float64 as one vector, this path 1.38e-3 (fp32) -- the same
# distance from the truth, median per-tensor ratio 1.08.  The float32 ORACLE alone is a poor yardstick (1.6e-4 on this step: its
# evaluation order happens to flip few PReLU elements; it was the only yardstick before round 4).  PReLU-flip noise is an independent,
# heavy-tailed sample in each float32 evaluation of a tensor, so:
#   * whole vector, where it averages out: at most 2 x the worse of the two float32 yardsticks (fp32); bf16x3 rounds its forward to 1e-5
#     instead of 1e-6, ten times the flips, sqrt(10) = 3.2 x the deviation (measured 3.8 x): factor 6;
#   * per tensor: 6 x (10 x) the worse yardstick of THAT tensor, with a floor of 2 x (6 x) the reference's whole-vector deviation for
#     tensors where both yardsticks happen to be flip-free (e.g. decoders.4: 4e-6).  Measured above the floor: ratio <= 3.2 (5.1).
YARD = {"fp32": (6.0, 2.0, 2.0), "bf16x3": (10.0, 6.0, 6.0)}


def ops_precision():
    import importlib
    return importlib.import_module("i-dccrn-vae_amd").ops.PRECISION


def _mi_fixture(golden):
    d = golden("op_mi")
    return d, (lambda k: torch.from_numpy(d[k]).cuda()), int(d["ns"])


def test_mutual_information_reference(golden):
    """complex_standard_vae_loss.mutual_information (model/pretrain_pvaes_loss.py:129-159; idv_mi_fwd / idv_mi_bwd): value and
    the gradients to miu, log_sigma, delta and the samples against the reference class's autograd (fixture op_mi.npz; two
    components sit inside the |delta| >= sigma guard)."""
    pl = importlib.import_module("i-dccrn-vae_amd.model.pretrain_pvaes_loss")
    d, T_, ns = _mi_fixture(golden)
    loss = pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.7, 'multiple', 'real_imag', [1, 1, 0], ns)
    with torch.no_grad():
        assert abs(float(loss.mutual_information(T_("miu"), T_("log_sigma"), T_("delta"), T_("z"))) - float(d["mi"])) < 2e-5
    leaves = [T_(k).requires_grad_(True) for k in ("miu", "log_sigma", "delta", "z")]
    mi = loss.mutual_information(*leaves)
    assert abs(float(mi.detach()) - float(d["mi"])) < 2e-5
    (3.0 * mi).backward()
    for n, t in zip(("miu", "log_sigma", "delta", "z"), leaves):
        e = relerr(t.grad.cpu() / 3.0, torch.from_numpy(d[f"g_{n}"]))
        assert e < 2e-4, (n, e)


@pytest.mark.parametrize("recon", ["multiple", "prob"])
@pytest.mark.parametrize("prior", ["ri_inde", "ri_corr"])
def test_elbo_branches_reference(golden, recon, prior):
    """cal_loss (model/pretrain_pvaes_loss.py:313-347) with mi_weight 0.7 in the four (recon_loss_type, prior_mode) branches:
    the seven returned values and the gradients of the final loss to the posterior and the samples, against the reference."""
    pl = importlib.import_module("i-dccrn-vae_amd.model.pretrain_pvaes_loss")
    d, T_, ns = _mi_fixture(golden)
    loss = pl.complex_standard_vae_loss(torch.ones(1), 0.8, 0.7, recon, 'real_imag', [1.0, 0.5, 0.25], ns, prior)
    leaves = [T_(k).requires_grad_(True) for k in ("miu", "log_sigma", "delta", "z")]
    stft_rep = T_("stft_source").repeat_interleave(ns, dim=0)
    pred = torch.view_as_complex(T_("pred").contiguous())
    res = loss.cal_loss(T_("source"), T_("est"), stft_rep, pred, *leaves, 5)
    want = d[f"loss:{recon}:{prior}"]
    for k, (a, b) in enumerate(zip(res, want)):
        assert abs(float(a) - float(b)) <= 1e-4 * max(1.0, abs(float(b))), (k, float(a), float(b))
    res[0].backward()
    for n, t in zip(("miu", "log_sigma", "delta", "z"), leaves):
        e = relerr(t.grad.cpu(), torch.from_numpy(d[f"g:{recon}:{prior}:{n}"]))
        assert e < 5e-4, (n, e)


def test_grad_cvae_mi_reference(pm, losses, golden):
    """pretrained_vaes/train.py with --mi_weight != 0: the CVAE step of test_grad_cvae_reference with the mutual-information
    term on (pretrain_pvaes_loss.py:334-343), on the encoder's own planar posterior and samples; the seven returned values and
    the encoder's gradients against the reference (fixture grad_cvae_mi_mini.npz: reconstruction weights 0.01 so that the KL
    and MI terms carry a visible share of the gradient)."""
    d = golden("grad_cvae_mi_mini")
    _, plm, _ = losses
    base, seed, zdim, ns = int(d["base"]), int(d["seed"]), int(d["zdim"]), int(d["ns"])
    np_ = O.net_params(True, base)
    enc = load_synth(pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns), seed)
    dec = load_synth(pm.pvae_dccrn_decoder_skip_prepare(np_, True, "cuda", ns, zdim, NFFT, HOP, WIN, "real_imag", SKIP), seed + 1)
    x = T_(d["x"]).cuda()
    B, L = x.shape
    eps = (T_(d["eps_r"]).cuda(), T_(d["eps_i"]).cuda())
    w = [float(v) for v in d["weights"]]
    with torch.enable_grad():
        z, miu, ls, dl, skiper, C, F, stft_x = enc(x, train=True, eps=eps)
        recon, pred = dec(stft_x, z, skiper, C, F, train=True)
        xr = x.unsqueeze(1).repeat(1, ns, 1).view(B * ns, L)
        sx = stft_x.detach().unsqueeze(1).repeat(1, ns, 1, 1, 1).view(B * ns, stft_x.shape[1], stft_x.shape[2], 2)
        pl = plm.complex_standard_vae_loss(torch.ones(1), 1.0, float(d["mi_weight"]), 'multiple', 'real_imag', w, ns)
        lo = pl.cal_loss(xr, recon, sx, pred, miu, ls, dl, z, 5)
        lo[0].backward()
    for k, (a, b) in enumerate(zip(lo, T_(d["loss"]))):
        assert abs(float(a.detach()) - float(b)) < 3e-4 * max(1.0, abs(float(b))), (k, float(a.detach()), float(b))
    print("worst", check_grads(d, "enc.", enc))

