"""Parity AT THE BASELINE BATCH SIZES: the launches whose throughput bench.py reports - DCCRN-CL B = 64,
CVAE B = 64 x 5 samples (320 decoder rows: planar offsets beyond 2^31 floats), NSVAE / two-phase B = 32 x 2 samples - compared
with the REAL reference's outputs.  The committed full-width fixtures (tests/golden/make_golden.py full / vaefull, written by
the imported reference) hold B = 2 or B = 1 x 2 samples; their inputs and injected eps are tiled to the benchmark batch, so every
utterance row must reproduce its fixture row and every loss (a mean over rows that repeat the fixture's rows equally often)
must reproduce the fixture's value.  Eval-mode batch norm: rows are independent, exactly as in the reference."""
import importlib

import numpy as np
import pytest
import torch

from oracle import idccrn_oracle as O
from conftest import relerr

pytestmark = pytest.mark.gpu
NFFT, HOP, WIN = 512, 100, 400
SKIP = [0, 1, 2, 3, 4, 5]


@pytest.fixture(scope="module")
def pm():
    return importlib.import_module("i-dccrn-vae_amd.model.pvae_module")


@pytest.fixture(scope="module")
def losses():
    return (importlib.import_module("i-dccrn-vae_amd.model.nsvae_loss"),
            importlib.import_module("i-dccrn-vae_amd.model.pretrain_pvaes_loss"))


def T_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_synth(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(O.synth_state_dict(shapes, seed))
    return module.cuda()


def rows_match(got, want_rows, row_of, tol, what):
    """got [R, ...] on the GPU; row r must equal want_rows[row_of(r)] (relative L2 per row)."""
    want_rows = want_rows.cuda().double()
    worst = 0.0
    for r in range(got.shape[0]):
        w = want_rows[row_of(r)]
        e = float((got[r].double() - w).norm() / (w.norm() + 1e-30))
        worst = max(worst, e)
        assert e < tol, (what, r, e)
    return worst


def test_dccrn_cl_batch_64(pm, losses, golden):
    """The headline launch shapes: B = 64 4-s utterances (J = 64 * 642 columns) against the reference's waveform."""
    d = golden("dccrn_full_eval")
    B, rep = 64, 32
    np_ = O.net_params(True, int(d["base"]))
    m = load_synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None), int(d["seed"]))
    x = T_(d["x"]).repeat(rep, 1).cuda()
    assert x.shape[0] == B
    with torch.no_grad():
        est, predict = m(x, train=False)
        assert tuple(est.shape) == (B, 64000)
        print("worst row", rows_match(est, T_(d["clean"]), lambda r: r % 2, 1e-4, "waveform"))
        pr = torch.view_as_real(predict)[:, ::8, ::16]
        rows_match(pr, T_(d["pred_sub"]), lambda r: r % 2, 1e-4, "predict")
        rows_match(m.std_DCCRN.latent, T_(d["latent"]), lambda r: r % 2, 1e-4, "latent")
        nl, _ = losses
        clean_ref = T_(d["clean_ref"]).repeat(rep, 1).cuda()
        got = nl.ete_train_se_loss([0.0, 0.0, 1.0]).final_ete_loss(predict, m.stft(clean_ref), clean_ref, est)
        for a, b in zip(got, T_(d["loss"])):
            assert abs(float(a) - float(b)) < 2e-4 * max(1.0, abs(float(b)))


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_cvae_batch_64_by_5_samples(pm, losses, golden, precision):
    """BASELINE config 2 at its batch: B = 64, num_samples = 5 -> 320 decoder rows.  The fixture has one utterance with two
    samples (eps_0, eps_1); here sample s of utterance b takes eps_{(b + s) mod 2}, so row (b, s) must equal fixture row
    (b + s) mod 2 and both fixture rows occur 160 times: the ELBO terms (means over rows) equal the fixture's."""
    ops = pm.ops
    nl, pl = losses
    dc = golden("vae_cvae_full_eval")
    base, seed, zdim = int(dc["base"]), int(dc["seed"]), int(dc["zdim"])
    assert int(dc["ns"]) == 2
    B, ns = 64, 5
    tol = 1e-4 if precision == "fp32" else 1e-3
    np_ = O.net_params(True, base)
    x1 = T_(dc["x"])
    L = x1.shape[1]
    T = 1 + L // HOP
    rng = lambda sd_, *shape: torch.from_numpy(np.random.default_rng(sd_).standard_normal(shape).astype("float32"))
    e_r, e_i = rng(seed + 300, 1, 2, T, zdim)[0], rng(seed + 301, 1, 2, T, zdim)[0]       # the fixture's eps: [2, T, zdim]
    pick = torch.tensor([[(b + s) % 2 for s in range(ns)] for b in range(B)])              # [B, ns]
    eps = (e_r[pick].cuda(), e_i[pick].cuda())                                            # [B, ns, T, zdim]
    row_of = lambda r: (r // ns + r % ns) % 2
    keep = ops.PRECISION
    try:
        ops.set_precision(precision)
        with torch.no_grad():
            enc = load_synth(pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns), seed)
            dec = load_synth(pm.pvae_dccrn_decoder_skip_prepare(np_, True, "cuda", ns, zdim, NFFT, HOP, WIN, "real_imag", SKIP), seed + 1)
            x = x1.repeat(B, 1).cuda()
            z, miu, ls, dl, skiper, C, F, stft_x = enc(x, train=False, eps=eps)
            assert tuple(z.shape)[0] == B * ns
            rows_match(z[:, ::8, ::4], T_(dc["z_sub"]), row_of, tol, "z")
            rows_match(miu[:, ::8, ::4], T_(dc["miu_sub"]), lambda r: 0, tol, "miu")
            rows_match(ls[:, ::8, ::4], T_(dc["ls_sub"]), lambda r: 0, tol, "log_sigma")
            rows_match(dl[:, ::8, ::4], T_(dc["dl_sub"]), lambda r: 0, tol, "delta")
            rows_match(skiper[5][:, ::8, :, ::8], T_(dc["skip5_sub"]), lambda r: 0, tol, "skip5")
            recon, predict = dec(stft_x, z, skiper, C, F, train=False)
            assert tuple(recon.shape) == (B * ns, L)
            print("worst row", rows_match(recon[:, ::16], T_(dc["recon_sub"]), row_of, tol, "recon"))
            rows_match(torch.view_as_real(predict)[:, ::8, ::16], T_(dc["pred_sub"]), row_of, tol, "predict")
            xr, sx = x.repeat_interleave(ns, dim=0), stft_x.repeat_interleave(ns, dim=0)
            loss = pl.complex_standard_vae_loss(torch.ones(1), 1.0, 0.0, 'multiple', 'real_imag', [1.0, 1.0, 0.0], ns)
            out = loss.cal_loss(xr, recon, sx, predict, miu, ls, dl, z, 5)
            for a, b in zip([out[0], out[1], out[2], out[4], out[5], out[6]], T_(dc["elbo"])):
                assert abs(float(a) - float(b)) < 10 * tol * max(1.0, abs(float(b))), (float(a), float(b))
    finally:
        ops.set_precision(keep)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_nsvae_twophase_batch_32_by_2_samples(pm, losses, golden, precision):
    """BASELINE configs 3 / 5 at their batch: NSVAE encoder (H = 768) + fine-tuned decoder with the real skips repeated per
    sample (pad='sig', mask), B = 32 x 2 samples; every row against the reference's B = 1 x 2 fixture."""
    ops = pm.ops
    nl, pl = losses
    dc, dn = golden("vae_cvae_full_eval"), golden("vae_nsvae_full_eval")
    base, seed, zdim, ns = int(dn["base"]), int(dn["seed"]), int(dn["zdim"]), int(dn["ns"])
    B = 32
    tol = 1e-4 if precision == "fp32" else 1e-3
    np_ = O.net_params(True, base)
    x1 = T_(dn["x"])
    L = x1.shape[1]
    T = 1 + L // HOP
    rng = lambda sd_, *shape: torch.from_numpy(np.random.default_rng(sd_).standard_normal(shape).astype("float32"))
    keep = ops.PRECISION
    try:
        ops.set_precision(precision)
        with torch.no_grad():
            x = x1.repeat(B, 1).cuda()
            enc2 = load_synth(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns, 2), seed + 2)
            dec2 = load_synth(pm.nsvae_pvae_dccrn_decoder_twophase(np_, True, "cuda", ns, zdim, NFFT, HOP, WIN, "mask", True, SKIP, False), seed + 3)
            eps2 = tuple(rng(seed + 310 + k, 1, ns, T, zdim).repeat(B, 1, 1, 1).cuda() for k in range(4))
            z_s, miu_s, ls_s, dl_s, z_n, miu_n, ls_n, dl_n, skiper2, C, F, stft_x2 = enc2(x, train=False, eps=eps2)
            row_of = lambda r: r % ns
            for got, name, ro in ((z_s, "z_speech_sub", row_of), (z_n, "z_noise_sub", row_of), (miu_s, "miu_speech_sub", lambda r: 0),
                                  (miu_n, "miu_noise_sub", lambda r: 0), (ls_s, "ls_speech_sub", lambda r: 0),
                                  (dl_n, "dl_noise_sub", lambda r: 0)):
                rows_match(got[:, ::8, ::4], T_(dn[name]), ro, tol, name)
            recon2, pred2 = dec2(stft_x2, z_s, skiper2, C, F, train=False, pad='sig')
            assert tuple(recon2.shape) == (B * ns, L)
            print("worst row", rows_match(recon2[:, ::16], T_(dn["recon_sub"]), row_of, tol, "recon"))
            rows_match(torch.view_as_real(pred2)[:, ::8, ::16], T_(dn["pred_sub"]), row_of, tol, "predict")
            xr, sx = x.repeat_interleave(ns, dim=0), stft_x2.repeat_interleave(ns, dim=0)
            got = nl.two_phase_loss([0, 0, 1], 1.0, zdim, 1).phase_2_loss(pred2, sx, xr, recon2, None, None, None, None)
            for a, b in zip(got[:4], T_(dn["phase2"])):
                assert abs(float(a) - float(b)) < 10 * tol * max(1.0, abs(float(b))), (float(a), float(b))
            # nsvae KL of the noisy encoder's latents against the CVAE encoder's (the fixture's pairing), same batch
            enc = load_synth(pm.pvae_dccrn_encoder_skip_prepare(np_, True, "cuda", zdim, NFFT, HOP, WIN, ns), seed)
            e1 = (rng(seed + 300, 1, ns, T, zdim).repeat(B, 1, 1, 1).cuda(), rng(seed + 301, 1, ns, T, zdim).repeat(B, 1, 1, 1).cuda())
            z, miu, ls, dl = enc(x, train=False, eps=e1)[:4]
            rows_match(z[:, ::8, ::4], T_(dc["z_sub"]), row_of, tol, "cvae z")
            L_ = nl.standard_nsvae_loss_true_kl(1.0, 0, 1.0, 0.0, zdim, ns, 2, 'original', 'False', [], 'both')
            o2 = L_.final_nsvae_loss(miu, miu, miu_s, miu_n, ls, ls, ls_s, ls_n, dl, dl, dl_s, dl_n, z_s, z_n, None, None, None)
            for a, b in zip(o2[:4], T_(dn["nsvae"])):
                assert abs(float(a) - float(b)) < 10 * tol * max(1.0, abs(float(b))), (float(a), float(b))
    finally:
        ops.set_precision(keep)
