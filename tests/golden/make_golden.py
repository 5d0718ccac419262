#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REAL REFERENCE.

Run in the build container only (the reference lives at /root/reference and never
travels):  ``PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py``

Only inputs and expected outputs are stored (data, no reference source).  Weights are
not stored: both sides rebuild them from ``oracle.idccrn_oracle.synth_tensor(name,
shape, seed)``.  While generating, every oracle function is checked against the
reference output, so a fixture is only written when the oracle is pinned.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("IDCCRN_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import idccrn_oracle as O  # noqa: E402

import model.complex_progress as R_cp  # noqa: E402  (reference)
import model.pvae_module as R_pm  # noqa: E402  (reference)
import model.sisnr_loss as R_sisnr  # noqa: E402
import model.nsvae_loss as R_nl  # noqa: E402
import model.pretrain_pvaes_loss as R_pl  # noqa: E402
import model.causal_netconfig as R_cnc  # noqa: E402
import model.net_config as R_nc  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)
NFFT, HOP, WIN = 512, 100, 400


def rnd(seed, *shape, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((scale * g.standard_normal(shape)).astype("float32"))


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(name, got, want, tol=2e-5):
    e = relerr(got, want)
    status = "ok" if e <= tol else "FAIL"
    print(f"  [{status}] oracle vs reference  {name:40s} rel={e:.2e}")
    assert e <= tol, name


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def load_synth(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = O.synth_state_dict(shapes, seed)
    module.load_state_dict(sd, strict=True)
    module.eval()
    return sd


def fix_bn_flags(module, flag):
    for m in module.modules():
        if isinstance(m, R_cp.ComplexBatchNormal):
            m.init_flag = flag


# ----------------------------------------------------------------------------- ops
def gen_ops():
    print("== unit fixtures")
    # STFT / ISTFT
    x = rnd(1, 2, 3200, scale=0.1)
    st = R_pm.STFT(NFFT, HOP, WIN, "cpu")
    ist = R_pm.ISTFT(NFFT, HOP, WIN, "cpu")
    X = st(x)
    check("stft", O.stft(x, NFFT, HOP, WIN), X)
    Y = rnd(2, 2, 257, 33, 2)
    y = ist(torch.complex(Y[..., 0], Y[..., 1]))
    check("istft", O.istft(Y, NFFT, HOP, WIN), y)
    save("op_stft", x=x, X=X, Y=Y, y=y)

    # complex conv (causal and not), complex transposed conv
    for causal in (True, False):
        cin, cout, F, T = 3, 8, 17, 9
        xin = rnd(3, 2, cin, F, T, 2)
        pad = (2, 1) if causal else (2, 0)
        cls = R_cp.causal_complex_conv2d if causal else R_cp.ComplexConv2d
        m = cls(cin, cout, (5, 2), (2, 1), pad)
        sd = load_synth(m, 11)
        y = m(xin)
        args = (sd["conv_re.weight"], sd["conv_re.bias"], sd["conv_im.weight"], sd["conv_im.bias"], (2, 1), pad, causal)
        check(f"cconv causal={causal}", O.complex_conv2d(xin, *args), y)
        check(f"cconv naive causal={causal}", O.complex_conv2d_naive(xin, *args), y)
        save(f"op_cconv_{'causal' if causal else 'plain'}", x=xin, y=y, cin=cin, cout=cout, seed=11)

        cin, cout, F, T = 6, 4, 9, 9
        xin = rnd(4, 2, cin, F, T, 2)
        cls = R_cp.causal_ComplexConvTranspose2d if causal else R_cp.ComplexConvTranspose2d
        m = cls(cin, cout, (5, 2), (2, 1), (2, 0))
        sd = load_synth(m, 12)
        y = m(xin)
        args = (sd["tconv_re.weight"], sd["tconv_re.bias"], sd["tconv_im.weight"], sd["tconv_im.bias"], (2, 1), (2, 0), causal)
        check(f"cconvT causal={causal}", O.complex_conv_transpose2d(xin, *args), y)
        check(f"cconvT naive causal={causal}", O.complex_conv_transpose2d_naive(xin, *args), y)
        save(f"op_cconvt_{'causal' if causal else 'plain'}", x=xin, y=y, cin=cin, cout=cout, seed=12)

    # complex batch norm: train (first call + momentum call) and eval
    C = 8
    xin = rnd(5, 3, C, 9, 7, 2) * 1.5 + 0.3
    bn = R_cp.ComplexBatchNormal(C, 0, 0)
    sd = load_synth(bn, 13)
    y_eval = bn(xin, train=False)
    g = lambda k: sd[k]
    check("cbn eval", O.cbn_whiten_affine(xin, g("running_mean_real"), g("running_mean_imag"), g("Vrr"), g("Vri"), g("Vii"),
                                          g("gamma_rr"), g("gamma_ri"), g("gamma_ii"), g("beta_r"), g("beta_i")), y_eval)
    bn.init_flag = True
    y_train = bn(xin, train=True)
    stats = O.cbn_batch_stats(xin)
    check("cbn train", O.cbn_whiten_affine(xin, *stats, g("gamma_rr"), g("gamma_ri"), g("gamma_ii"), g("beta_r"), g("beta_i")), y_train)
    first = {k: bn.state_dict()[k].clone() for k in ("running_mean_real", "running_mean_imag", "Vrr", "Vri", "Vii")}
    for k, s in zip(("running_mean_real", "running_mean_imag", "Vrr", "Vri", "Vii"), stats):
        check(f"cbn running[{k}] first call", s, first[k])
    xin2 = rnd(6, 3, C, 9, 7, 2) * 0.7 - 0.2
    bn(xin2, train=True)
    second = {k: bn.state_dict()[k].clone() for k in first}
    stats2 = O.cbn_batch_stats(xin2)
    for k, s1, s2 in zip(first, stats, stats2):
        check(f"cbn running[{k}] momentum", 0.9 * s1 + 0.1 * s2, second[k])
    save("op_cbn", x=xin, x2=xin2, y_eval=y_eval, y_train=y_train, seed=13, C=C,
         **{"first_" + k: v for k, v in first.items()}, **{"second_" + k: v for k, v in second.items()})

    # PReLU
    pr = torch.nn.PReLU()
    pr.weight.fill_(0.2)
    check("prelu", O.prelu(xin, pr.weight), pr(xin))

    # complex LSTM
    T, B, I, H = 7, 3, 20, 16
    xin = rnd(7, T, B, I, 2)
    m = R_cp.ComplexLSTM(I, H, "cpu", num_layers=2)
    sd = load_synth(m, 14)
    y = m(xin)
    check("complex lstm", O.complex_lstm(xin, sd, "", 2), y)
    save("op_clstm", x=xin, y=y, seed=14, I=I, H=H)

    # complex dense
    m = R_cp.ComplexDense(16, 40)
    sd = load_synth(m, 15)
    xin = rnd(8, 21, 16, 2)
    y = m(xin)
    check("complex dense", O.complex_dense(xin, sd["linear_read.weight"], sd["linear_read.bias"],
                                           sd["linear_imag.weight"], sd["linear_imag.bias"]), y)
    save("op_cdense", x=xin, y=y, seed=15)

    # SI-SNR known answer (model/sisnr_loss.py:27-30) and random
    src = torch.tensor([1, 2, 3, 4, 5, 1, 2, 3, 4, 5]).view(2, 5).float()
    est = torch.tensor([[1.5, 2.5, 3.5, 4.5, 5.5], [1.5, 2.5, 3.5, 4.5, 5.5]])
    ka = R_sisnr.si_snr(src, est)
    print("  si_snr known answer:", float(ka))
    check("si_snr known answer", O.si_snr(src, est), ka)
    check("si_snr matmul form", O.si_snr_matmul_form(src, est), ka)
    s2, e2 = rnd(9, 4, 1600, scale=0.1), rnd(10, 4, 1600, scale=0.1)
    e2 = s2 + 0.3 * e2
    r2 = R_sisnr.si_snr(s2, e2)
    check("si_snr random", O.si_snr(s2, e2), r2)
    save("op_sisnr", src=src, est=est, known=ka, s2=s2, e2=e2, r2=r2)

    # recon loss
    P = rnd(11, 3, 257, 9, 2)
    Or = rnd(12, 3, 257, 9, 2)
    so, es = rnd(13, 3, 800, scale=0.1), rnd(14, 3, 800, scale=0.1)
    L = R_nl.ete_train_se_loss([0.3, 0.5, 1.0])
    want = L.final_ete_loss(torch.complex(P[..., 0], P[..., 1]), Or, so, es)
    got = O.multiple_recon_loss(P, Or, so, es, [0.3, 0.5, 1.0])
    for n, a, b in zip(("final", "cpx", "mag", "sisnr"), got, want):
        check("recon loss " + n, a, b)
    save("op_recon", P=P, O=Or, source=so, est=es, want=torch.stack(list(want)))

    # reparameterisation (randn_like injected) and KLs
    B, T, H, ns = 2, 5, 8, 3
    miu, ls, dl = rnd(15, B, T, H, 2), rnd(16, B, T, H, 2, scale=0.3), rnd(17, B, T, H, 2, scale=0.8)
    eps_r, eps_i = rnd(18, B, ns, T, H), rnd(19, B, ns, T, H)
    enc = R_pm.pvae_dccrn_encoder_skip_prepare(O.net_params(True, 4), True, "cpu", H, NFFT, HOP, WIN, ns)
    draws = [eps_r, eps_i]
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: draws.pop(0)
    try:
        z = enc.reparameterization(miu, ls, dl, ns)
    finally:
        torch.randn_like = orig
    check("reparameterization", O.reparameterization(miu, ls, dl, ns, eps_r, eps_i), z)
    miu2, ls2, dl2 = rnd(20, B, T, H, 2), rnd(21, B, T, H, 2, scale=0.3), rnd(22, B, T, H, 2, scale=0.8)
    pl = R_pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', [1, 1, 0], ns)
    kl_p = pl.cal_kl_arbi_prior(miu, miu2, ls, ls2, dl, dl2)
    check("kl pretrain", O.complex_kl(miu, miu2, ls, ls2, dl, dl2, 1e-9).mean(), kl_p)
    nl = R_nl.standard_nsvae_loss_true_kl(1.0, 0, 1.0, 0.5, H, ns, 2, 'original', 'False', [], 'both')
    kl_n = nl.cal_kl(miu, miu2, ls, ls2, dl, dl2, None)
    check("kl nsvae", O.complex_kl(miu, miu2, ls, ls2, dl, dl2, 1e-10), kl_n)
    miu3, ls3, dl3 = rnd(23, B, T, H, 2), rnd(24, B, T, H, 2, scale=0.3), rnd(25, B, T, H, 2, scale=0.8)
    miu4, ls4, dl4 = rnd(26, B, T, H, 2), rnd(27, B, T, H, 2, scale=0.3), rnd(28, B, T, H, 2, scale=0.8)
    want = nl.final_nsvae_loss(miu, miu2, miu3, miu4, ls, ls2, ls3, ls4, dl, dl2, dl3, dl4, None, None, None, None, None)
    got = O.nsvae_loss(miu, miu2, miu3, miu4, ls, ls2, ls3, ls4, dl, dl2, dl3, dl4, 1.0, 1.0, 0.5, 2)
    for n, a, b in zip(("final", "kl", "kl_clean", "kl_noise", "dis_s", "dis_n"), got, want[:6]):
        check("nsvae loss " + n, a, b)
    save("op_vae", miu=miu, ls=ls, dl=dl, miu2=miu2, ls2=ls2, dl2=dl2, miu3=miu3, ls3=ls3, dl3=dl3,
         miu4=miu4, ls4=ls4, dl4=dl4, eps_r=eps_r, eps_i=eps_i, z=z, kl_pretrain=kl_p, kl_nsvae=kl_n,
         nsvae=torch.stack([torch.as_tensor(w).float() for w in want[:6]]))


# ----------------------------------------------------------------------------- models
def gen_dccrn(tag, base, B, L, seed, train, causal=True, full_outputs=True):
    print(f"== DCCRN_ {tag}: base={base} B={B} L={L} train={train} causal={causal}")
    np_ref = (R_cnc if causal else R_nc).get_net_params()
    np_ = O.net_params(causal, base)
    if base == 32:
        for k in ("encoder_channels", "decoder_channels", "lstm_dim", "dense", "encoder_paddings"):
            assert [tuple(v) if isinstance(v, (list, tuple)) else v for v in np_ref[k]] == \
                   [tuple(v) if isinstance(v, (list, tuple)) else v for v in np_[k]], k
    skip = [0, 1, 2, 3, 4, 5]
    m = R_pm.DCCRN_(NFFT, HOP, np_, causal, "cpu", WIN, skip, "mask", False, None, None)
    sd = load_synth(m, seed)
    x = rnd(seed + 100, B, L, scale=0.1)
    fix_bn_flags(m, True)
    clean, pred = m(x, train=train)
    bs = O.BNState()
    o_clean, o_pred, o_lat = O.dccrn_forward(x, sd, np_, causal, NFFT, HOP, WIN, skip, "mask", train, bs)
    check("waveform", o_clean, clean, 1e-4)
    check("predict stft", o_pred, torch.view_as_real(pred), 1e-4)
    out = dict(x=x, clean=clean, seed=seed, base=base, train=int(train), causal=int(causal))
    pr = torch.view_as_real(pred)
    if full_outputs:
        out["pred"] = pr
    else:
        out["pred_sub"] = pr[:, ::8, ::16]
        out["pred_l2"] = pr.double().norm()
        out["pred_mean"] = pr.double().mean()
    if train:
        after = m.state_dict()
        for k in after:
            if k.endswith(("running_mean_real", "running_mean_imag", "Vrr", "Vri", "Vii")) and ".bn." in k:
                out["bn:" + k] = after[k]
    else:
        out["latent"] = m.std_DCCRN.latent
        check("latent", o_lat, m.std_DCCRN.latent, 1e-4)
    clean_ref = rnd(seed + 200, B, clean.shape[1], scale=0.1)
    L_ = R_nl.ete_train_se_loss([0.0, 0.0, 1.0])
    loss = L_.final_ete_loss(pred, m.stft(clean_ref), clean_ref, clean)
    out["clean_ref"] = clean_ref
    out["loss"] = torch.stack(list(loss))
    save(f"dccrn_{tag}", **out)


def gen_vae(tag, base, zdim, B, L, ns, seed, train):
    print(f"== VAE {tag}: base={base} zdim={zdim} B={B} L={L} ns={ns} train={train}")
    np_ = O.net_params(True, base)
    skip = [0, 1, 2, 3, 4, 5]
    x = rnd(seed + 100, B, L, scale=0.1)
    T = 1 + L // HOP
    # ---- CVAE encoder/decoder (skip_prepare), real_imag
    enc = R_pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns)
    dec = R_pm.pvae_dccrn_decoder_skip_prepare(np_, True, "cpu", ns, zdim, NFFT, HOP, WIN, "real_imag", skip)
    sd_e, sd_d = load_synth(enc, seed), load_synth(dec, seed + 1)
    eps = [rnd(seed + 300, B, ns, T, zdim), rnd(seed + 301, B, ns, T, zdim)]
    draws = list(eps)
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: draws.pop(0)
    try:
        fix_bn_flags(enc, True)
        z, miu, ls, dl, skiper, C, F, stft_x = enc(x, train=train)
    finally:
        torch.randn_like = orig
    fix_bn_flags(dec, True)
    recon, pred = dec(stft_x, z, skiper, C, F, train=train)
    oe = O.vae_encoder_forward(x, sd_e, np_, True, zdim, NFFT, HOP, WIN, ns, 1, eps, train)
    check("cvae z", oe["z_speech"], z, 1e-4)
    check("cvae miu", oe["miu_speech"], miu, 1e-4)
    o_rec, o_pred = O.vae_decoder_forward(oe["stft_x"], oe["z_speech"], oe["skiper"], C, F, sd_d, np_, True, ns,
                                          NFFT, HOP, WIN, "real_imag", skip, "zero", True, train)
    check("cvae recon", o_rec, recon, 1e-4)
    xr = x[:, :recon.shape[1]].repeat_interleave(ns, dim=0)
    sx = stft_x.repeat_interleave(ns, dim=0)
    pl = R_pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', [1.0, 1.0, 0.0], ns)
    lo = pl.cal_loss(xr, recon, sx, pred, miu, ls, dl, z, 5)
    ol = O.cvae_elbo(xr, o_rec, sx, o_pred, oe["miu_speech"], oe["log_sigma_speech"], oe["delta_speech"], 1.0, [1.0, 1.0, 0.0])
    check("cvae elbo", ol[0], lo[0], 1e-4)
    check("cvae kl", ol[2], lo[2], 1e-4)
    save(f"vae_cvae_{tag}", x=x, eps_r=eps[0], eps_i=eps[1], z=z, miu=miu, log_sigma=ls, delta=dl,
         skip5=skiper[5], skip0_sub=skiper[0][:, ::4, ::8, ::4], recon=recon, pred_sub=torch.view_as_real(pred)[:, ::4, ::4],
         elbo=torch.stack([torch.as_tensor(v).float() for v in (lo[0], lo[1], lo[2], lo[4], lo[5], lo[6])]),
         seed=seed, base=base, zdim=zdim, ns=ns, train=int(train))

    # ---- NSVAE twophase encoder (latent_num=2) + twophase decoder (mask, pad='sig')
    enc2 = R_pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns, 2)
    dec2 = R_pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, "cpu", ns, zdim, NFFT, HOP, WIN, "mask", True, skip, False)
    sd_e2, sd_d2 = load_synth(enc2, seed + 2), load_synth(dec2, seed + 3)
    eps2 = [rnd(seed + 310 + i, B, ns, T, zdim) for i in range(4)]
    draws = list(eps2)
    torch.randn_like = lambda t, *a, **k: draws.pop(0)
    try:
        fix_bn_flags(enc2, True)
        r = enc2(x, train=train)
    finally:
        torch.randn_like = orig
    z_s, miu_s, ls_s, dl_s, z_n, miu_n, ls_n, dl_n, skiper2, C, F, stft_x2 = r
    fix_bn_flags(dec2, True)
    recon2, pred2 = dec2(stft_x2, z_s, skiper2, C, F, train=train, pad='sig')
    oe2 = O.vae_encoder_forward(x, sd_e2, np_, True, zdim, NFFT, HOP, WIN, ns, 2, eps2, train)
    check("nsvae z_speech", oe2["z_speech"], z_s, 1e-4)
    check("nsvae z_noise", oe2["z_noise"], z_n, 1e-4)
    o_rec2, o_pred2 = O.vae_decoder_forward(oe2["stft_x"], oe2["z_speech"], oe2["skiper"], C, F, sd_d2, np_, True, ns,
                                            NFFT, HOP, WIN, "mask", skip, "sig", True, train)
    check("twophase recon", o_rec2, recon2, 1e-4)
    tl = R_nl.two_phase_loss([0, 0, 1], 1.0, zdim, 1)
    l2 = tl.phase_2_loss(pred2, sx, xr, recon2, None, None, None, None)
    save(f"vae_nsvae_{tag}", x=x, **{f"eps{i}": e for i, e in enumerate(eps2)}, z_speech=z_s, z_noise=z_n,
         miu_speech=miu_s, miu_noise=miu_n, log_sigma_speech=ls_s, log_sigma_noise=ls_n, delta_speech=dl_s,
         delta_noise=dl_n, recon=recon2, pred_sub=torch.view_as_real(pred2)[:, ::4, ::4],
         phase2=torch.stack([torch.as_tensor(v).float() for v in l2[:4]]),
         seed=seed, base=base, zdim=zdim, ns=ns, train=int(train))


# ----------------------------------------------------------------------------- gradients (reference autograd)
def summarize(t, limit=16384, cap=8192):
    """Full tensor when small, otherwise a strided subsample; the consumer re-derives the stride from the shapes."""
    t = t.detach().reshape(-1)
    if t.numel() <= limit:
        return t.clone()
    stride = -(-t.numel() // cap)
    return t[::stride].clone()


def grad_record(out, prefix, module, limit=16384, cap=8192):
    """Per parameter: summarised gradient + its L2 norm (norm catches errors outside the subsample)."""
    for k, p_ in module.named_parameters():
        if p_.grad is None:
            continue
        out[f"g:{prefix}{k}"] = summarize(p_.grad, limit, cap)
        out[f"n:{prefix}{k}"] = p_.grad.double().norm()


def adam_record(out, prefix, module, lr=1e-3, wd=1e-3):
    """One Adam step as the reference trainers take it (supervised_dccrn/train.py:109, :243): updated weights."""
    params = [p_ for p_ in module.parameters() if p_.grad is not None]
    opt = torch.optim.Adam(params, lr=lr, weight_decay=wd)
    opt.step()
    for k, p_ in module.named_parameters():
        if p_.grad is not None and (k.endswith("conv_re.weight") or k.endswith("gamma_ri") or k.endswith("prelu.weight")
                                    or k.endswith("weight_hh_l1") or k.endswith("linear_read.weight")
                                    or k.endswith("tconv_im.weight")):
            out[f"a:{prefix}{k}"] = summarize(p_)


def gen_grad_dccrn(tag, base, B, L, seed, weights, limit=16384, cap=8192, adam=True):
    print(f"== grads DCCRN_ {tag}: base={base} B={B} L={L} weights={weights}")
    np_ = O.net_params(True, base)
    skip = [0, 1, 2, 3, 4, 5]
    with torch.enable_grad():
        m = R_pm.DCCRN_(NFFT, HOP, np_, True, "cpu", WIN, skip, "mask", False, None, None)
        load_synth(m, seed)
        m.train()
        fix_bn_flags(m, True)
        x = rnd(seed + 100, B, L, scale=0.1).requires_grad_(True)
        clean_ref = rnd(seed + 200, B, L, scale=0.1)
        est, pred = m(x, train=True)
        loss = R_nl.ete_train_se_loss(weights).final_ete_loss(pred, m.stft(clean_ref), clean_ref, est)
        loss[0].backward()
    out = dict(x=x.detach(), clean_ref=clean_ref, seed=seed, base=base, weights=np.asarray(weights, dtype="float32"),
               loss=torch.stack([v.detach() for v in loss]), gx=x.grad, est=est.detach(), sum_limit=limit, sum_cap=cap)
    grad_record(out, "", m, limit, cap)
    if adam:
        with torch.no_grad():
            adam_record(out, "", m)
    save(f"grad_dccrn_{tag}", **out)


def gen_grad_vae(tag, base, zdim, B, L, ns, seed, limit=16384, cap=8192, adam=True):
    """The three VAE train steps against the reference's own loss.backward().  tag "mini": reduced width, every gradient element
    (up to `limit`) + one Adam step; tag "full": the reference's full width (base 32, zdim 128 -> LSTM hidden 384 / 768, K = 1280
    projections, repeated-skip decoder), gradients subsampled to `cap` elements per tensor + full L2 norms."""
    print(f"== grads VAE {tag}: base={base} zdim={zdim} B={B} L={L} ns={ns}")
    np_ = O.net_params(True, base)
    skip = [0, 1, 2, 3, 4, 5]
    T = 1 + L // HOP
    orig = torch.randn_like
    # ---- CVAE pre-training step (pretrained_vaes/train.py:281-301), recon weights incl. SI-SNR
    w = [1.0, 1.0, 0.3]
    with torch.enable_grad():
        enc = R_pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns)
        dec = R_pm.pvae_dccrn_decoder_skip_prepare(np_, True, "cpu", ns, zdim, NFFT, HOP, WIN, "real_imag", skip)
        load_synth(enc, seed), load_synth(dec, seed + 1)
        enc.train(), dec.train()
        fix_bn_flags(enc, True), fix_bn_flags(dec, True)
        x = rnd(seed + 100, B, L, scale=0.1)
        eps = [rnd(seed + 300, B, ns, T, zdim), rnd(seed + 301, B, ns, T, zdim)]
        draws = list(eps)
        torch.randn_like = lambda t, *a, **k: draws.pop(0)
        try:
            z, miu, ls, dl, skiper, C, F, stft_x = enc(x, train=True)
        finally:
            torch.randn_like = orig
        recon, pred = dec(stft_x, z, skiper, C, F, train=True)
        xr = x.unsqueeze(1).repeat(1, ns, 1).view(B * ns, L)
        sx = stft_x.unsqueeze(1).repeat(1, ns, 1, 1, 1).view(B * ns, stft_x.shape[1], stft_x.shape[2], 2)
        pl = R_pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', w, ns)
        lo = pl.cal_loss(xr, recon, sx, pred, miu, ls, dl, z, 5)
        lo[0].backward()
    out = dict(x=x, eps_r=eps[0], eps_i=eps[1], seed=seed, base=base, zdim=zdim, ns=ns, weights=np.asarray(w, dtype="float32"),
               loss=torch.stack([torch.as_tensor(v).detach().float() for v in (lo[0], lo[1], lo[2], lo[4], lo[5], lo[6])]))
    out.update(sum_limit=limit, sum_cap=cap)
    grad_record(out, "enc.", enc, limit, cap)
    grad_record(out, "dec.", dec, limit, cap)
    if adam:
        with torch.no_grad():
            adam_record(out, "enc.", enc), adam_record(out, "dec.", dec)
    save(f"grad_cvae_{tag}", **out)

    # ---- NSVAE step (train_nsvae.py:487-574): frozen clean / noise encoders (eval, no_grad), noisy encoder trains
    with torch.enable_grad():
        ce = R_pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns)
        ne = R_pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns)
        ye = R_pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns, 2)
        load_synth(ce, seed + 4), load_synth(ne, seed + 5), load_synth(ye, seed + 6)
        ye.train()
        fix_bn_flags(ye, True)
        clean = rnd(seed + 110, B, L, scale=0.1)
        noise = rnd(seed + 111, B, L, scale=0.1)
        noisy = clean + noise
        epsn = [rnd(seed + 320 + i, B, ns, T, zdim) for i in range(8)]
        draws = list(epsn)
        torch.randn_like = lambda t, *a, **k: draws.pop(0)
        try:
            with torch.no_grad():
                zc, mc, lc, dc, skc, _, _, _ = ce(clean, train=False)
                zn, mn, ln_, dn, skn, _, _, _ = ne(noise, train=False)
            r = ye(noisy, train=True)
        finally:
            torch.randn_like = orig
        zs, ms, ls_, ds, znn, mnn, lnn, dnn, sky, C, F, stft_y = r
        nl = R_nl.standard_nsvae_loss_true_kl(1.0, 0, 1.0, 0.5, zdim, ns, 2, 'original', 'False', [], 'both')
        ln_out = nl.final_nsvae_loss(mc, mn, ms, mnn, lc, ln_, ls_, lnn, dc, dn, ds, dnn, zs, znn, skc, skn, sky)
        ln_out[0].backward()
    out = dict(clean=clean, noise=noise, seed=seed, base=base, zdim=zdim, ns=ns,
               **{f"eps{i}": e for i, e in enumerate(epsn)},
               loss=torch.stack([torch.as_tensor(v).detach().float() for v in ln_out[:6]]))
    out.update(sum_limit=limit, sum_cap=cap)
    grad_record(out, "noisy.", ye, limit, cap)
    if adam:
        with torch.no_grad():
            adam_record(out, "noisy.", ye)
    save(f"grad_nsvae_{tag}", **out)

    # ---- decoder fine-tune step (train_second_phase_decoder.py:376-433): frozen noisy encoder (eval), decoder trains
    with torch.enable_grad():
        ye2 = R_pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns, 2)
        de2 = R_pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, "cpu", ns, zdim, NFFT, HOP, WIN, "mask", True, skip, False)
        load_synth(ye2, seed + 6), load_synth(de2, seed + 7)
        for p_ in ye2.parameters():
            p_.requires_grad = False
        de2.train()
        fix_bn_flags(de2, True)
        eps2 = [rnd(seed + 340 + i, B, ns, T, zdim) for i in range(4)]
        draws = list(eps2)
        torch.randn_like = lambda t, *a, **k: draws.pop(0)
        try:
            r = ye2(noisy, train=False)
        finally:
            torch.randn_like = orig
        zs, _, _, _, _, _, _, _, sky, C, F, stft_y = r
        rec, prd = de2(stft_y, zs, sky, C, F, train=True, pad='sig')
        sxc = ye2.stft(clean).unsqueeze(1).repeat(1, ns, 1, 1, 1).view(B * ns, stft_y.shape[1], stft_y.shape[2], 2)
        cb = clean.unsqueeze(1).repeat(1, ns, 1).view(B * ns, L)
        tl = R_nl.two_phase_loss([0, 0, 1], 1.0, zdim, 1)
        l2 = tl.phase_2_loss(prd, sxc, cb, rec, None, None, None, None)
        l2[0].backward()
    out = dict(clean=clean, noise=noise, seed=seed, base=base, zdim=zdim, ns=ns,
               **{f"eps{i}": e for i, e in enumerate(eps2)},
               loss=torch.stack([torch.as_tensor(v).detach().float() for v in l2[:4]]), recon=rec.detach())
    out.update(sum_limit=limit, sum_cap=cap)
    grad_record(out, "dec.", de2, limit, cap)
    if adam:
        with torch.no_grad():
            adam_record(out, "dec.", de2)
    save(f"grad_twophase_{tag}", **out)


def gen_grad_cvae_mi(tag, base, zdim, B, L, ns, seed, mi_weight=2.0):
    """The CVAE pre-training step of gen_grad_vae with the mutual-information term on (pretrained_vaes/train.py --mi_weight;
    pretrain_pvaes_loss.py:334-343): loss values and the ENCODER's gradients (the decoder's do not see the term)."""
    print(f"== grads CVAE + MI {tag}: mi_weight={mi_weight}")
    np_ = O.net_params(True, base)
    T = 1 + L // HOP
    orig = torch.randn_like
    w = [0.01, 0.01, 0.0]            # small reconstruction weights: the KL and MI terms carry a visible share of the gradient
    with torch.enable_grad():
        enc = R_pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns)
        dec = R_pm.pvae_dccrn_decoder_skip_prepare(np_, True, "cpu", ns, zdim, NFFT, HOP, WIN, "real_imag", [0, 1, 2, 3, 4, 5])
        load_synth(enc, seed), load_synth(dec, seed + 1)
        enc.train(), dec.train()
        fix_bn_flags(enc, True), fix_bn_flags(dec, True)
        x = rnd(seed + 100, B, L, scale=0.1)
        eps = [rnd(seed + 300, B, ns, T, zdim), rnd(seed + 301, B, ns, T, zdim)]
        draws = list(eps)
        torch.randn_like = lambda t, *a, **k: draws.pop(0)
        try:
            z, miu, ls, dl, skiper, C, F, stft_x = enc(x, train=True)
        finally:
            torch.randn_like = orig
        recon, pred = dec(stft_x, z, skiper, C, F, train=True)
        xr = x.unsqueeze(1).repeat(1, ns, 1).view(B * ns, L)
        sx = stft_x.unsqueeze(1).repeat(1, ns, 1, 1, 1).view(B * ns, stft_x.shape[1], stft_x.shape[2], 2)
        pl = R_pl.complex_standard_vae_loss(torch.ones(1), 1.0, mi_weight, 'multiple', 'real_imag', w, ns)
        lo = pl.cal_loss(xr, recon, sx, pred, miu, ls, dl, z, 5)
        lo[0].backward()
    out = dict(x=x, eps_r=eps[0], eps_i=eps[1], seed=seed, base=base, zdim=zdim, ns=ns, weights=np.asarray(w, dtype="float32"),
               mi_weight=np.asarray(mi_weight), loss=torch.stack([torch.as_tensor(v).detach().float() for v in lo]))
    print(f"  final {float(lo[0]):.5f} kl {float(lo[2]):.5f} mi {float(lo[3]):.5f}")
    grad_record(out, "enc.", enc)
    save(f"grad_cvae_mi_{tag}", **out)


def sub(t, *steps):
    idx = tuple(slice(None, None, s) for s in steps)
    return t[idx].clone()


def gen_vae_full(tag, B, ns, seed):
    """Full-width VAE workloads (base 32, zdim 128 -> LSTM hidden 384 / 768) in eval mode, 4 s utterances; outputs stored
    subsampled (+ norms) to keep the fixture small."""
    base, zdim, L = 32, 128, 64000
    print(f"== VAE full {tag}: B={B} ns={ns}")
    np_ = O.net_params(True, base)
    skip = [0, 1, 2, 3, 4, 5]
    T = 1 + L // HOP
    x = rnd(seed + 100, B, L, scale=0.1)
    orig = torch.randn_like
    enc = R_pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns)
    dec = R_pm.pvae_dccrn_decoder_skip_prepare(np_, True, "cpu", ns, zdim, NFFT, HOP, WIN, "real_imag", skip)
    load_synth(enc, seed), load_synth(dec, seed + 1)
    eps = [rnd(seed + 300, B, ns, T, zdim), rnd(seed + 301, B, ns, T, zdim)]
    draws = list(eps)
    torch.randn_like = lambda t, *a, **k: draws.pop(0)
    try:
        z, miu, ls, dl, skiper, C, F, stft_x = enc(x, train=False)
    finally:
        torch.randn_like = orig
    recon, pred = dec(stft_x, z, skiper, C, F, train=False)
    xr = x.repeat_interleave(ns, dim=0)
    sx = stft_x.repeat_interleave(ns, dim=0)
    pl = R_pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', [1.0, 1.0, 0.0], ns)
    lo = pl.cal_loss(xr, recon, sx, pred, miu, ls, dl, z, 5)
    save(f"vae_cvae_{tag}", x=x, seed=seed, base=base, zdim=zdim, ns=ns, train=0,
         z_sub=sub(z, 1, 8, 4, 1), z_l2=z.double().norm(), miu_sub=sub(miu, 1, 8, 4, 1), miu_l2=miu.double().norm(),
         ls_sub=sub(ls, 1, 8, 4, 1), dl_sub=sub(dl, 1, 8, 4, 1), skip5_sub=sub(skiper[5], 1, 8, 1, 8, 1),
         recon_sub=sub(recon, 1, 16), recon_l2=recon.double().norm(),
         pred_sub=sub(torch.view_as_real(pred), 1, 8, 16, 1), pred_l2=torch.view_as_real(pred).double().norm(),
         elbo=torch.stack([torch.as_tensor(v).float() for v in (lo[0], lo[1], lo[2], lo[4], lo[5], lo[6])]))

    enc2 = R_pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cpu", zdim, NFFT, HOP, WIN, ns, 2)
    dec2 = R_pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, "cpu", ns, zdim, NFFT, HOP, WIN, "mask", True, skip, False)
    load_synth(enc2, seed + 2), load_synth(dec2, seed + 3)
    eps2 = [rnd(seed + 310 + i, B, ns, T, zdim) for i in range(4)]
    draws = list(eps2)
    torch.randn_like = lambda t, *a, **k: draws.pop(0)
    try:
        r = enc2(x, train=False)
    finally:
        torch.randn_like = orig
    z_s, miu_s, ls_s, dl_s, z_n, miu_n, ls_n, dl_n, skiper2, C, F, stft_x2 = r
    recon2, pred2 = dec2(stft_x2, z_s, skiper2, C, F, train=False, pad='sig')
    tl = R_nl.two_phase_loss([0, 0, 1], 1.0, zdim, 1)
    l2 = tl.phase_2_loss(pred2, sx, xr, recon2, None, None, None, None)
    # nsvae loss of the noisy-encoder posteriors against the CVAE posterior above (clean = noise encoder here: data only)
    nl = R_nl.standard_nsvae_loss_true_kl(1.0, 0, 1.0, 0.0, zdim, ns, 2, 'original', 'False', [], 'both')
    ln_out = nl.final_nsvae_loss(miu, miu, miu_s, miu_n, ls, ls, ls_s, ls_n, dl, dl, dl_s, dl_n, z_s, z_n, None, None, None)
    save(f"vae_nsvae_{tag}", x=x, seed=seed, base=base, zdim=zdim, ns=ns, train=0,
         z_speech_sub=sub(z_s, 1, 8, 4, 1), z_speech_l2=z_s.double().norm(), z_noise_sub=sub(z_n, 1, 8, 4, 1),
         miu_speech_sub=sub(miu_s, 1, 8, 4, 1), miu_noise_sub=sub(miu_n, 1, 8, 4, 1), miu_noise_l2=miu_n.double().norm(),
         ls_speech_sub=sub(ls_s, 1, 8, 4, 1), dl_noise_sub=sub(dl_n, 1, 8, 4, 1),
         recon_sub=sub(recon2, 1, 16), recon_l2=recon2.double().norm(),
         pred_sub=sub(torch.view_as_real(pred2), 1, 8, 16, 1),
         phase2=torch.stack([torch.as_tensor(v).float() for v in l2[:4]]),
         nsvae=torch.stack([torch.as_tensor(v).float() for v in ln_out[:4]]))


def gen_datanorm():
    """DCCRN_ with data_mean / data_std (pvae_module.py:217-221, :235-238), mask and real_imag outputs, eval mode."""
    print("== DCCRN_ datanorm")
    np_ = O.net_params(True, 4)
    skip = [0, 1, 2, 3, 4, 5]
    mean = rnd(91, 1, 257, 1, 2, scale=0.05)
    std = rnd(92, 1, 257, 1, 2, scale=0.2).abs() + 0.5
    x = rnd(93, 2, 1600, scale=0.1)
    out = dict(x=x, data_mean=mean, data_std=std, seed=94)
    for rt in ("mask", "real_imag"):
        m = R_pm.DCCRN_(NFFT, HOP, np_, True, "cpu", WIN, skip, rt, False, mean, std)
        sd = O.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if v is not None and k not in ("data_mean", "data_std")}, 94)
        sd["data_mean"], sd["data_std"] = mean, std
        m.load_state_dict(sd, strict=True)
        m.eval()
        clean, pred = m(x, train=False)
        out[f"clean_{rt}"] = clean
        out[f"pred_{rt}"] = torch.view_as_real(pred)
        o_clean, o_pred, _ = O.dccrn_forward(x, sd, np_, True, NFFT, HOP, WIN, skip, rt, False, None, mean, std)
        check(f"datanorm {rt} waveform", o_clean, clean, 1e-4)
        check(f"datanorm {rt} predict", o_pred, torch.view_as_real(pred), 1e-4)
    # the same model as a TRAIN step (the reference's loss.backward() with --data_norm): loss + parameter gradients
    with torch.enable_grad():
        m = R_pm.DCCRN_(NFFT, HOP, np_, True, "cpu", WIN, skip, "mask", False, mean, std)
        sd = O.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if v is not None and k not in ("data_mean", "data_std")}, 94)
        sd["data_mean"], sd["data_std"] = mean, std
        m.load_state_dict(sd, strict=True)
        m.train()
        fix_bn_flags(m, True)
        clean_ref = rnd(95, 2, 1600, scale=0.1)
        w = [0.2, 0.1, 1.0]
        est, pred = m(x, train=True)
        loss = R_nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(clean_ref), clean_ref, est)
        loss[0].backward()
    out.update(train_clean_ref=clean_ref, train_weights=np.asarray(w, dtype="float32"),
               train_loss=torch.stack([v.detach() for v in loss]), train_est=est.detach())
    grad_record(out, "train:", m)
    save("dccrn_datanorm_mini", **out)


def gen_checkpoint():
    """A tiny run folder written by the REFERENCE classes exactly as supervised_dccrn/train.py:295-324 writes it
    (state_dict file + checkpoint dict with optimizer / scheduler state), plus the reference's eval output of the
    stored weights.  Data only: tensors and bookkeeping scalars."""
    import shutil
    print("== checkpoint fixture")
    np_ = O.net_params(True, 2, 16)
    skip = [0, 1, 2, 3, 4, 5]
    folder = os.path.join(HERE, "ckpt", "2025-01-01-00h00_DCCRN_causal=True_skipuse=012345_reconw=001_recontype=mask_resynthesis=False_datanorm=False")
    shutil.rmtree(os.path.join(HERE, "ckpt"), ignore_errors=True)
    os.makedirs(folder)
    with torch.enable_grad():
        m = R_pm.DCCRN_(NFFT, HOP, np_, True, "cpu", WIN, skip, "mask", False, None, None)
        load_synth(m, 97)
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0.001)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, 'min', factor=0.5, patience=3)
        x = rnd(98, 2, 1600, scale=0.1)
        c = rnd(99, 2, 1600, scale=0.1)
        est, pred = m(x, train=True)
        loss = R_nl.ete_train_se_loss([0.0, 0.0, 1.0]).final_ete_loss(pred, m.stft(c), c, est)[0]
        opt.zero_grad()
        loss.backward()
        opt.step()
        sch.step(float(loss))
    torch.save(m.state_dict(), os.path.join(folder, "DCCRN_curr_best_epoch.pt"))
    torch.save({"epoch": 3, "best_val_loss": float(loss), "cpt_patience": 1, "model_state_dict": m.state_dict(),
                "loss_log": {"train_loss": np.asarray([3.0, 2.0, 1.5, 1.2]), "val_loss": np.asarray([3.1, 2.2, 1.6, 1.3])},
                "model_optim_dict": opt.state_dict(), "model_scheduler_dict": sch.state_dict()},
               os.path.join(folder, "DCCRN_checkpoint.pt"))
    with torch.no_grad():
        m.eval()
        xe = rnd(100, 2, 1600, scale=0.1)
        clean, _ = m(xe, train=False)
    np.savez_compressed(os.path.join(folder, "expected.npz"), x=xe.numpy(), clean=clean.numpy(),
                        step1=np.asarray(opt.state_dict()["state"][0]["step"]))
    print("wrote", folder, sum(os.path.getsize(os.path.join(folder, f)) for f in os.listdir(folder)) // 1024, "KiB")


def _extract_functions(path, names):
    """The named top-level function definitions of a reference file that cannot be imported here (its module-level imports
    need librosa / soundfile / pesq), compiled from the file's own syntax tree and run as they stand: no stand-in modules,
    nothing of the reference is written anywhere."""
    import ast
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), [n.name for n in picked]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def gen_evalpath():
    """the three `outtype` estimators of the evaluation script and its SI-SDR metric, from the reference's own
    function bodies (test_se_cvaefinetune.py:85-135, utils/eval_metrics.py:49-64)."""
    print("== evaluation-path estimators")
    import contextlib
    import io
    rim, cm, psm = _extract_functions(os.path.join(REF, "i_dccrn_vae", "nsvae_dccrn", "test_se_cvaefinetune.py"),
                                      ["real_and_imag_mask", "complex_mask", "phase_sensitive_mask"])
    (sisdr,) = _extract_functions(os.path.join(REF, "utils", "eval_metrics.py"), ["compute_sisdr"])
    ns_, F, T = 3, 17, 23
    speech = torch.complex(rnd(401, ns_, F, T), rnd(402, ns_, F, T))
    noise = torch.complex(rnd(403, ns_, F, T, scale=0.7), rnd(404, ns_, F, T, scale=0.7))
    noisy = rnd(405, 1, F, T, 2)
    out = {}
    for name, ref_fn, ora in (("real_imag_mask", rim, O.outtype_real_imag_mask), ("complex_mask", cm, O.outtype_complex_mask),
                              ("phase_mask", psm, O.outtype_phase_sensitive_mask)):
        with contextlib.redirect_stdout(io.StringIO()):          # complex_mask prints a debug line
            want = ref_fn(noise.clone(), speech.clone(), noisy.clone())
        got = ora(noise, speech, noisy)
        check("outtype " + name, torch.view_as_real(got), torch.view_as_real(want), 1e-6)
        out[name] = torch.view_as_real(want)
    save("op_outtype", speech=torch.view_as_real(speech), noise=torch.view_as_real(noise), noisy=noisy, **out)
    est = rnd(406, 4000, scale=0.1).numpy()
    ref = (torch.from_numpy(est) * 0.8 + rnd(407, 4000, scale=0.03)).numpy()
    want = float(sisdr(est, ref))
    got = float(O.sisdr_np(est, ref))
    print(f"  sisdr reference {want:.6f} oracle {got:.6f}")
    assert abs(want - got) < 1e-5
    save("op_sisdr", est=est, ref=ref, sisdr=np.float64(want))


def gen_resi():
    """residual_loss of the REAL standard_nsvae_loss_true_kl (model/nsvae_loss.py:363-446) in its five modes."""
    print("== residual (skip-matching) loss")
    B, F, T = 2, 5, 7
    chans = [4, 6, 8]
    clean = [rnd(500 + i, B, c, F, T, 2) for i, c in enumerate(chans)]
    noise = [rnd(510 + i, B, c, F, T, 2) for i, c in enumerate(chans)]
    noisy1 = [rnd(520 + i, B, c, F, T, 2) for i, c in enumerate(chans)]          # same width as the clean encoder
    noisy2 = [rnd(530 + i, B, 2 * c, F, T, 2) for i, c in enumerate(chans)]      # speech | noise halves
    out = dict(chans=np.asarray(chans), skip_to_use=np.asarray([0, 2]))
    for i in range(3):
        out.update({f"clean{i}": clean[i], f"noise{i}": noise[i], f"noisy1_{i}": noisy1[i], f"noisy2_{i}": noisy2[i]})
    modes = {"l1_plain": (1, "original", "speech", noisy1), "l1_split": (1, "adapt", "speech", noisy2),
             "l2_both": (2, "original", "both", noisy2), "l2_speech_split": (2, "double", "speech", noisy2),
             "l2_speech_plain": (2, "original", "speech", noisy1)}
    for name, (latent_num, model, matching, noisy) in modes.items():
        L_ = R_nl.standard_nsvae_loss_true_kl(1.0, 0.5, 1.0, 0.0, 16, 2, latent_num, model, "True", [0, 2], matching)
        want = L_.residual_loss(clean, noise, noisy)
        got = O.residual_loss(clean, noise, noisy, [0, 2], latent_num, model in ("adapt", "double"), matching)
        for a, b in zip(got, want):
            assert abs(float(a) - float(b)) <= 1e-6 * max(1.0, abs(float(b))), (name, float(a), float(b))
        print(f"  [ok] oracle vs reference  residual_loss {name}: {[round(float(v), 6) for v in want]}")
        out[name] = np.asarray([float(v) for v in want], dtype="float64")
    save("op_resi", **out)


@torch.enable_grad()
def gen_mi():
    """mutual_information / cal_gaussian_prob / the 'prob' and 'ri_corr' branches of the REAL complex_standard_vae_loss
    (model/pretrain_pvaes_loss.py:64-182, :313-347), values and autograd gradients."""
    print("== CVAE ELBO: mutual information, prob recon, ri_corr prior")
    B, ns, T, H, F = 3, 2, 5, 6, 7
    miu = rnd(600, B, T, H, 2) * 0.5
    log_sigma = rnd(601, B, T, H, 2) * 0.3
    delta = rnd(602, B, T, H, 2) * 0.3
    delta[0, 0, :2] = torch.tensor([1.4, -0.9])                       # |delta| > sigma: the 0.90 guard fires
    eps_r, eps_i = rnd(603, B, ns, T, H), rnd(604, B, ns, T, H)
    z = O.reparameterization(miu, log_sigma, delta, ns, eps_r, eps_i)
    z = z + 0.3 * rnd(605, *z.shape)
    out = dict(miu=miu, log_sigma=log_sigma, delta=delta, z=z, ns=np.asarray(ns))
    leaves = [t.clone().requires_grad_(True) for t in (miu, log_sigma, delta, z)]
    pl = R_pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.7, 'multiple', 'real_imag', [1, 1, 0], ns)
    lp = pl.cal_gaussian_prob(miu, log_sigma, delta, z.view(B, ns, T, H, 2))
    check("gaussian_logprob", O.gaussian_logprob(miu, log_sigma, delta, z.view(B, ns, T, H, 2)), lp, 1e-5)
    mi = pl.mutual_information(*leaves)
    grads = torch.autograd.grad(mi, leaves)
    oleaves = [t.clone().requires_grad_(True) for t in (miu, log_sigma, delta, z)]
    omi = O.mutual_information(*oleaves, ns)
    ograds = torch.autograd.grad(omi, oleaves)
    assert abs(float(omi) - float(mi)) < 1e-5 * max(1.0, abs(float(mi)))
    for n, a, b in zip(("miu", "log_sigma", "delta", "z"), ograds, grads):
        check(f"mutual_information d/d{n}", a, b, 1e-4)
        out[f"g_{n}"] = b
    out["mi"] = np.asarray(float(mi))
    out["logprob"] = lp
    print(f"  mutual information {float(mi):.6f}")
    # cal_loss in the four (recon, prior) branches with the MI term on
    src, est = rnd(610, B * ns, 400) * 0.1, rnd(611, B * ns, 400) * 0.1
    stft_src = rnd(612, B, F, T, 2)
    pred = torch.view_as_complex(rnd(613, B * ns, F, T, 2).contiguous())
    out.update(source=src, est=est, stft_source=stft_src, pred=torch.view_as_real(pred))
    stft_rep = stft_src.repeat_interleave(ns, dim=0)
    for recon in ("multiple", "prob"):
        for prior in ("ri_inde", "ri_corr"):
            pl = R_pl.complex_standard_vae_loss(torch.ones(1), 0.8, 0.7, recon, 'real_imag', [1.0, 0.5, 0.25], ns, prior)
            leaves = [t.clone().requires_grad_(True) for t in (miu, log_sigma, delta, z)]
            res = pl.cal_loss(src, est, stft_rep, pred, *leaves, 5)
            want = [float(v) for v in res]
            got = O.cvae_elbo_full(src, est, stft_rep, torch.view_as_real(pred), miu, log_sigma, delta, z, 0.8, 0.7, recon,
                                   [1.0, 0.5, 0.25], ns, prior)
            order = (0, 1, 2, 3, 4, 5, 6)
            for k in order:
                assert abs(float(got[k]) - want[k]) <= 2e-5 * max(1.0, abs(want[k])), (recon, prior, k, float(got[k]), want[k])
            g = torch.autograd.grad(res[0], leaves)
            out[f"loss:{recon}:{prior}"] = np.asarray(want, dtype="float64")
            for n, t in zip(("miu", "log_sigma", "delta", "z"), g):
                out[f"g:{recon}:{prior}:{n}"] = t
            print(f"  [ok] oracle vs reference  cal_loss {recon}/{prior}: final {want[0]:.6f} kl {want[2]:.6f} mi {want[3]:.6f}")
    save("op_mi", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "mini", "full"]
    if "evalpath" in which:
        gen_evalpath()
    if "resi" in which:
        gen_resi()
    if "mi" in which:
        gen_mi()
        gen_grad_cvae_mi("mini", 4, 32, 3, 1600, 2, 71)
    if "ops" in which:
        gen_ops()
    if "mini" in which:
        gen_dccrn("mini_eval", 4, 2, 1600, 21, train=False)
        gen_dccrn("mini_train", 4, 2, 1600, 22, train=True)
        gen_dccrn("mini_noncausal_eval", 4, 2, 1600, 23, train=False, causal=False)
        gen_vae("mini_eval", 4, 16, 2, 1600, 3, 31, train=False)
        gen_vae("mini_train", 4, 16, 2, 1600, 2, 41, train=True)
    if "full" in which:
        torch.set_num_threads(os.cpu_count())
        gen_dccrn("full_eval", 32, 2, 64000, 51, train=False, full_outputs=False)
    if "grads" in which:
        gen_grad_dccrn("mini", 4, 2, 1600, 61, [0.2, 0.1, 1.0])
        gen_grad_vae("mini", 4, 32, 2, 1600, 2, 71)
    if "gradfull" in which:
        # the supervised train step at the reference's FULL width (base 32), 1 s utterances: gradients subsampled to 1024
        # elements per tensor + full L2 norms
        torch.set_num_threads(os.cpu_count())
        gen_grad_dccrn("full", 32, 2, 16000, 91, [0.2, 0.1, 1.0], limit=2048, cap=1024, adam=False)
    if "gradvaefull" in which:
        # the three VAE train steps (BASELINE configs 2-5) at the reference's FULL width: base 32, zdim 128 (LSTM hidden 384 for the
        # CVAE encoder, 768 for the noisy encoder), 1 s utterances, B = 2, ns = 2
        torch.set_num_threads(os.cpu_count())
        gen_grad_vae("full", 32, 128, 2, 16000, 2, 93, limit=2048, cap=1024, adam=False)
    if "extras" in which or "datanorm" in which:
        gen_datanorm()
    if "extras" in which:
        gen_checkpoint()
    if "vaefull" in which:
        torch.set_num_threads(os.cpu_count())
        gen_vae_full("full_eval", 1, 2, 81)
