"""CPU oracle for the I-DCCRN-VAE enhancement hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``i-dccrn-vae_amd``) never routes through this
file and fails loudly when the HIP library is missing.

It is a plain restatement (torch CPU tensors used as an array library, fp32 or
fp64) of the algorithm of the reference repository for every row of
SURVEY.md section 8(a).  Each function cites the reference file:line whose behaviour
it restates.  Parity is PINNED: ``tests/golden/*.npz`` were produced by
importing the real reference modules in the build container
(``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py`` checks
every function below against them.

Tensor convention (same as the reference): complex feature maps are real
tensors ``[B, C, F, T, 2]`` with (real, imag) in the last dim.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as Fnn

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# STFT / ISTFT                                                                #
# --------------------------------------------------------------------------- #
def hann_periodic(win_length: int, dtype=torch.float32) -> Tensor:
    """torch.hann_window default (periodic=True), model/pvae_module.py:16."""
    n = torch.arange(win_length, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * n / win_length)).to(dtype)


def padded_window(n_fft: int, win_length: int, dtype=torch.float32) -> Tensor:
    """hann(win_length) centred inside an n_fft frame (torch.stft semantics)."""
    left = (n_fft - win_length) // 2
    w = torch.zeros(n_fft, dtype=dtype)
    w[left:left + win_length] = hann_periodic(win_length, dtype)
    return w


def stft(signal: Tensor, n_fft: int, hop: int, win_length: int) -> Tensor:
    """model/pvae_module.py:21-27 (torch.stft, center=True, reflect pad, onesided,
    unnormalised) -> [B, n_fft//2+1, 1 + L//hop, 2].

    X[f,t] = sum_n wp[n] * xp[hop*t + n] * exp(-2 pi i f n / n_fft)
    with xp = reflect-pad(x, n_fft//2).  Frame t is centred on sample hop*t.
    """
    x = signal
    B, L = x.shape
    half = n_fft // 2
    idx = torch.arange(-half, L + half)
    idx = idx.abs()                                   # reflect on the left edge
    idx = torch.where(idx >= L, 2 * (L - 1) - idx, idx)  # reflect on the right edge
    xp = x[:, idx]                                    # [B, L + n_fft]
    T = 1 + L // hop
    frames = xp.unfold(1, n_fft, hop)[:, :T]          # [B, T, n_fft]
    wp = padded_window(n_fft, win_length, x.dtype)
    spec = torch.fft.rfft(frames * wp, dim=-1)        # [B, T, F]
    spec = spec.transpose(1, 2)                       # [B, F, T]
    return torch.stack((spec.real, spec.imag), dim=-1).contiguous()


def istft(spec_ri: Tensor, n_fft: int, hop: int, win_length: int) -> Tensor:
    """model/pvae_module.py:38-42 (torch.istft, center=True, length=None).

    y = OLA_t( wp * irfft(X[:,t]) ) / OLA_t( wp^2 ), then drop n_fft//2 samples at
    both ends -> length hop*(T-1).  ``spec_ri`` is [B, F, T, 2].
    """
    B, F, T, _ = spec_ri.shape
    X = torch.complex(spec_ri[..., 0], spec_ri[..., 1]).transpose(1, 2)   # [B,T,F]
    fr = torch.fft.irfft(X, n=n_fft, dim=-1)                              # [B,T,n_fft]
    wp = padded_window(n_fft, win_length, fr.dtype)
    fr = fr * wp
    total = n_fft + hop * (T - 1)
    y = torch.zeros(B, total, dtype=fr.dtype)
    env = torch.zeros(total, dtype=fr.dtype)
    for t in range(T):
        y[:, t * hop:t * hop + n_fft] += fr[:, t]
        env[t * hop:t * hop + n_fft] += wp * wp
    half = n_fft // 2
    y = y[:, half:total - half]
    env = env[half:total - half]
    return y / env


# --------------------------------------------------------------------------- #
# complex layers (model/complex_progress.py)                                  #
# --------------------------------------------------------------------------- #
def complex_conv2d(x: Tensor, w_re: Tensor, b_re: Tensor, w_im: Tensor, b_im: Tensor,
                   stride: Tuple[int, int], padding: Tuple[int, int], causal: bool) -> Tensor:
    """model/complex_progress.py:32-36 (ComplexConv2d) and :16-22 (causal variant).

    re = conv_re(x_r) - conv_im(x_i);  im = conv_re(x_i) + conv_im(x_r); the causal
    variant drops the last time column after a symmetric time padding of 1.
    """
    xr, xi = x[..., 0], x[..., 1]
    re = Fnn.conv2d(xr, w_re, b_re, stride, padding) - Fnn.conv2d(xi, w_im, b_im, stride, padding)
    im = Fnn.conv2d(xi, w_re, b_re, stride, padding) + Fnn.conv2d(xr, w_im, b_im, stride, padding)
    if causal:
        re, im = re[:, :, :, :-1], im[:, :, :, :-1]
    return torch.stack((re, im), dim=-1)


def complex_conv2d_naive(x, w_re, b_re, w_im, b_im, stride, padding, causal):
    """Index-level restatement of complex_conv2d (loops over taps) used to check the
    library call above on small shapes; same reference lines."""
    B, Cin, Fi, Ti, _ = x.shape
    Co, _, KF, KT = w_re.shape
    sf, st = stride
    pf, pt = padding
    xp = torch.zeros(B, Cin, Fi + 2 * pf, Ti + 2 * pt, 2, dtype=x.dtype)
    xp[:, :, pf:pf + Fi, pt:pt + Ti] = x
    Fo = (Fi + 2 * pf - KF) // sf + 1
    To = (Ti + 2 * pt - KT) // st + 1
    re = torch.zeros(B, Co, Fo, To, dtype=x.dtype)
    im = torch.zeros(B, Co, Fo, To, dtype=x.dtype)
    for kf in range(KF):
        for kt in range(KT):
            patch = xp[:, :, kf:kf + sf * (Fo - 1) + 1:sf, kt:kt + st * (To - 1) + 1:st]  # [B,Cin,Fo,To,2]
            pr, pi = patch[..., 0], patch[..., 1]
            wr, wi = w_re[:, :, kf, kt], w_im[:, :, kf, kt]
            re += torch.einsum('oc,bcft->boft', wr, pr) - torch.einsum('oc,bcft->boft', wi, pi)
            im += torch.einsum('oc,bcft->boft', wr, pi) + torch.einsum('oc,bcft->boft', wi, pr)
    re += (b_re - b_im).view(1, -1, 1, 1)
    im += (b_re + b_im).view(1, -1, 1, 1)
    if causal:
        re, im = re[:, :, :, :-1], im[:, :, :, :-1]
    return torch.stack((re, im), dim=-1)


def complex_conv_transpose2d(x: Tensor, w_re: Tensor, b_re: Tensor, w_im: Tensor, b_im: Tensor,
                             stride, padding, causal: bool) -> Tensor:
    """model/complex_progress.py:275-279 (ComplexConvTranspose2d) and :244-250 (causal)."""
    xr, xi = x[..., 0], x[..., 1]
    re = Fnn.conv_transpose2d(xr, w_re, b_re, stride, padding) - Fnn.conv_transpose2d(xi, w_im, b_im, stride, padding)
    im = Fnn.conv_transpose2d(xi, w_re, b_re, stride, padding) + Fnn.conv_transpose2d(xr, w_im, b_im, stride, padding)
    if causal:
        re, im = re[:, :, :, :-1], im[:, :, :, :-1]
    return torch.stack((re, im), dim=-1)


def complex_conv_transpose2d_naive(x, w_re, b_re, w_im, b_im, stride, padding, causal):
    """Scatter-form restatement of the transposed convolution, same reference lines:
    out[fo=sf*fi-pf+kf, to=st*ti-pt+kt] += W[ci,co,kf,kt] * in[fi,ti]."""
    B, Cin, Fi, Ti, _ = x.shape
    _, Co, KF, KT = w_re.shape
    sf, st = stride
    pf, pt = padding
    Ffull = (Fi - 1) * sf + KF
    Tfull = (Ti - 1) * st + KT
    re = torch.zeros(B, Co, Ffull, Tfull, dtype=x.dtype)
    im = torch.zeros(B, Co, Ffull, Tfull, dtype=x.dtype)
    xr, xi = x[..., 0], x[..., 1]
    for kf in range(KF):
        for kt in range(KT):
            wr, wi = w_re[:, :, kf, kt], w_im[:, :, kf, kt]     # [Cin, Co]
            cr = torch.einsum('co,bcft->boft', wr, xr) - torch.einsum('co,bcft->boft', wi, xi)
            ci = torch.einsum('co,bcft->boft', wr, xi) + torch.einsum('co,bcft->boft', wi, xr)
            re[:, :, kf:kf + sf * (Fi - 1) + 1:sf, kt:kt + st * (Ti - 1) + 1:st] += cr
            im[:, :, kf:kf + sf * (Fi - 1) + 1:sf, kt:kt + st * (Ti - 1) + 1:st] += ci
    re = re[:, :, pf:Ffull - pf, pt:Tfull - pt] + (b_re - b_im).view(1, -1, 1, 1)
    im = im[:, :, pf:Ffull - pf, pt:Tfull - pt] + (b_re + b_im).view(1, -1, 1, 1)
    if causal:
        re, im = re[:, :, :, :-1], im[:, :, :, :-1]
    return torch.stack((re, im), dim=-1)


CBN_EPS = 1e-5


def cbn_batch_stats(x: Tensor):
    """model/complex_progress.py:131-143: per-channel mean and (co)variances over
    (B, F, T); Vrr and Vii carry +eps, Vri does not."""
    r, i = x[..., 0], x[..., 1]
    mu_r = r.mean(dim=(0, 2, 3), keepdim=True)
    mu_i = i.mean(dim=(0, 2, 3), keepdim=True)
    rc, ic = r - mu_r, i - mu_i
    Vrr = (rc * rc).mean(dim=(0, 2, 3), keepdim=True) + CBN_EPS
    Vii = (ic * ic).mean(dim=(0, 2, 3), keepdim=True) + CBN_EPS
    Vri = (rc * ic).mean(dim=(0, 2, 3), keepdim=True)
    return mu_r, mu_i, Vrr, Vri, Vii


def cbn_whiten_affine(x: Tensor, mu_r, mu_i, Vrr, Vri, Vii, g_rr, g_ri, g_ii, b_r, b_i) -> Tensor:
    """model/complex_progress.py:168-209 (cbn): 2x2 whitening then affine."""
    C = x.shape[1]
    v = lambda p: p.reshape(1, C, 1, 1)
    mu_r, mu_i, Vrr, Vri, Vii = map(v, (mu_r, mu_i, Vrr, Vri, Vii))
    g_rr, g_ri, g_ii, b_r, b_i = map(v, (g_rr, g_ri, g_ii, b_r, b_i))
    rc, ic = x[..., 0] - mu_r, x[..., 1] - mu_i
    delta = torch.clamp(Vrr * Vii - Vri * Vri + CBN_EPS, min=1e-8)
    s = torch.sqrt(delta)
    t = torch.sqrt(Vrr + Vii + 2 * s + CBN_EPS)
    inv = 1.0 / (s * t + CBN_EPS)
    Wrr, Wii, Wri = (Vii + s) * inv, (Vrr + s) * inv, -Vri * inv
    Zrr = g_rr * Wrr + g_ri * Wri
    Zri = g_rr * Wri + g_ri * Wii
    Zir = g_ri * Wrr + g_ii * Wri
    Zii = g_ri * Wri + g_ii * Wii
    out_r = Zrr * rc + Zri * ic + b_r
    out_i = Zir * rc + Zii * ic + b_i
    return torch.stack((out_r, out_i), dim=-1)


def prelu(x: Tensor, slope: Tensor) -> Tensor:
    """nn.PReLU() with one shared slope, model/pvae_module.py:58,82."""
    return torch.where(x >= 0, x, slope.reshape(()) * x)


def lstm_layer(x: Tensor, w_ih, w_hh, b_ih, b_hh) -> Tensor:
    """One unidirectional nn.LSTM layer, gate order (i, f, g, o), h0 = c0 = 0.
    x is [T, N, I] -> [T, N, H].  (torch.nn.LSTM as used at complex_progress.py:45-48.)"""
    T, N, _ = x.shape
    H = w_hh.shape[1]
    h = torch.zeros(N, H, dtype=x.dtype)
    c = torch.zeros(N, H, dtype=x.dtype)
    gx = x @ w_ih.t() + (b_ih + b_hh)
    out = []
    for t in range(T):
        g = gx[t] + h @ w_hh.t()
        i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out.append(h)
    return torch.stack(out, dim=0)


def lstm_stack(x: Tensor, sd: Dict[str, Tensor], prefix: str, num_layers: int) -> Tensor:
    for l in range(num_layers):
        x = lstm_layer(x, sd[f"{prefix}weight_ih_l{l}"], sd[f"{prefix}weight_hh_l{l}"],
                       sd[f"{prefix}bias_ih_l{l}"], sd[f"{prefix}bias_hh_l{l}"])
    return x


def complex_lstm(x: Tensor, sd: Dict[str, Tensor], prefix: str, num_layers: int = 2) -> Tensor:
    """model/complex_progress.py:50-74: four real LSTM passes;
    real = lstm_re(x_r) - lstm_im(x_i); imag = lstm_re(x_i) + lstm_im(x_r).  x is [T,B,I,2]."""
    xr, xi = x[..., 0], x[..., 1]
    rr = lstm_stack(xr, sd, prefix + "lstm_re.", num_layers)
    ri = lstm_stack(xr, sd, prefix + "lstm_im.", num_layers)
    ii = lstm_stack(xi, sd, prefix + "lstm_im.", num_layers)
    ir = lstm_stack(xi, sd, prefix + "lstm_re.", num_layers)
    return torch.stack((rr - ii, ir + ri), dim=-1)


def complex_dense(x: Tensor, w_r, b_r, w_i, b_i) -> Tensor:
    """model/complex_progress.py:83-89: two independent real linears (no cross terms)."""
    return torch.stack((x[..., 0] @ w_r.t() + b_r, x[..., 1] @ w_i.t() + b_i), dim=-1)


# --------------------------------------------------------------------------- #
# VAE pieces                                                                  #
# --------------------------------------------------------------------------- #
def guard_delta(sigma, d_r, d_i, eps):
    """|delta| <= sigma protection shared by reparameterisation (pvae_module.py:1840-1845)
    and the KL terms (pretrain_pvaes_loss.py:244-249, nsvae_loss.py:294-299)."""
    a = torch.sqrt(d_r * d_r + d_i * d_i + eps)
    scale = sigma * 0.99 / (a + eps)
    hit = a >= (sigma - 1e-3)
    return torch.where(hit, d_r * scale, d_r), torch.where(hit, d_i * scale, d_i)


def reparameterization(miu, log_sigma, delta, num_samples: int, eps_r: Tensor, eps_i: Tensor) -> Tensor:
    """model/pvae_module.py:1832-1886 with the two torch.randn_like draws injected
    (eps_r, eps_i are [B, num_samples, T, H]).  Returns [B*num_samples, T, H, 2]."""
    e = 1e-6
    mr, mi = miu[..., 0], miu[..., 1]
    sg = torch.exp(log_sigma[..., 0])
    dr, di = guard_delta(sg, delta[..., 0], delta[..., 1], e)
    a = torch.sqrt(dr * dr + di * di + e)
    den = torch.sqrt(2 * (sg + dr) + e)
    k_rr = (sg + dr) / (den + e)
    k_ir = di / (den + e)
    k_ii = torch.sqrt(sg * sg - a * a + e) / (den + e)
    u = lambda p: p.unsqueeze(1)
    zr = u(mr) + u(k_rr) * eps_r
    zi = u(mi) + u(k_ir) * eps_r + u(k_ii) * eps_i
    B, ns, T, H = zr.shape
    return torch.stack((zr.reshape(B * ns, T, H), zi.reshape(B * ns, T, H)), dim=-1)


def complex_kl(miu1, miu2, log_sigma1, log_sigma2, delta1, delta2, eps: float) -> Tensor:
    """Closed-form KL(q1 || q2) between improper complex Gaussians, per (b, t):
    model/pretrain_pvaes_loss.py:225-281 (eps=1e-9, caller takes the mean) and
    model/nsvae_loss.py:275-328 / :818-872 (eps=1e-10).  Returns [B, T]."""
    zdim = miu1.shape[2]
    s1, s2 = torch.exp(log_sigma1[..., 0]), torch.exp(log_sigma2[..., 0])
    d1r, d1i = guard_delta(s1, delta1[..., 0], delta1[..., 1], eps)
    d2r, d2i = guard_delta(s2, delta2[..., 0], delta2[..., 1], eps)
    a1 = d1r * d1r + d1i * d1i
    a2 = d2r * d2r + d2i * d2i
    logdet1 = torch.log(0.25 * (s1 * s1 - a1) + eps)
    logdet2 = torch.log(0.25 * (s2 * s2 - a2) + eps)
    coeff = 2.0 / (s2 * s2 - a2 + eps)
    trace = s1 * s2 - d2r * d1r - d2i * d1i
    dr = miu2[..., 0] - miu1[..., 0]
    di = miu2[..., 1] - miu1[..., 1]
    quad = dr * dr * (s2 - d2r) - 2 * d2i * dr * di + di * di * (s2 + d2r)
    return 0.5 * torch.sum(coeff * (trace + quad) + logdet2 - logdet1, dim=2) - zdim


# --------------------------------------------------------------------------- #
# losses                                                                      #
# --------------------------------------------------------------------------- #
def si_snr(source: Tensor, estimate: Tensor, eps: float = 1e-8) -> Tensor:
    """model/sisnr_loss.py:7-19 (copies at nsvae_loss.py:761-773, :877-889,
    pretrain_pvaes_loss.py:212-224).  The reference's BxB matmul + diag is the
    per-utterance dot product <est, s>; restated as three dot products."""
    energy = torch.sum(source * source, dim=1, keepdim=True)
    dot = torch.sum(estimate * source, dim=1, keepdim=True)
    s_t = dot * source / (energy + eps)
    e_n = estimate - s_t
    snr = 10 * torch.log10(torch.sum(s_t * s_t, dim=1) / (torch.sum(e_n * e_n, dim=1) + eps) + eps)
    return -snr.mean()


def si_snr_matmul_form(source: Tensor, estimate: Tensor, eps: float = 1e-8) -> Tensor:
    """Literal matmul/diagonal form of sisnr_loss.py:10-15, to show both forms agree."""
    B = source.shape[0]
    energy = torch.sum(source ** 2, dim=1).view(B, 1)
    d = torch.diag(torch.diagonal(estimate @ source.t(), 0))
    s_t = (d @ source) / (energy + eps)
    e_n = estimate - s_t
    snr = 10 * torch.log10(torch.sum(s_t ** 2, dim=1) / (torch.sum(e_n ** 2, dim=1) + eps) + eps)
    return -snr.mean()


def multiple_recon_loss(pred_ri: Tensor, ori_ri: Tensor, source: Tensor, est: Tensor,
                        weights: Sequence[float]):
    """model/nsvae_loss.py:775-797 (= :891-913, pretrain_pvaes_loss.py:184-206).
    pred_ri / ori_ri are [B, F, T, 2].  NOTE: the original magnitude uses the REAL part
    twice (nsvae_loss.py:783) - reproduced on purpose."""
    pr, pi = pred_ri[..., 0], pred_ri[..., 1]
    o_r, o_i = ori_ri[..., 0], ori_ri[..., 1]
    pmag = torch.sqrt(pr * pr + pi * pi + 1e-6)
    omag = torch.sqrt(o_r * o_r + o_r * o_r + 1e-6)
    l_cpx = (torch.sum((pr - o_r) ** 2, dim=1) + torch.sum((pi - o_i) ** 2, dim=1)).mean()
    l_mag = torch.sum((pmag - omag) ** 2, dim=1).mean()
    l_snr = si_snr(source, est)
    final = weights[0] * l_cpx + weights[1] * l_mag + weights[2] * l_snr
    return final, l_cpx, l_mag, l_snr


def cvae_elbo(source, est, stft_source, pred_ri, miu, log_sigma, delta, kl_weight: float,
              weights: Sequence[float]):
    """complex_standard_vae_loss.cal_loss with recon_loss_type='multiple', prior
    'ri_inde', mi_weight=0 (model/pretrain_pvaes_loss.py:313-347)."""
    recon, l_cpx, l_mag, l_snr = multiple_recon_loss(pred_ri, stft_source, source, est, weights)
    z = torch.zeros_like
    kl = complex_kl(miu, z(miu), log_sigma, z(log_sigma), delta, z(delta), 1e-9).mean()
    return recon + kl_weight * kl, recon, kl, l_cpx, l_mag, l_snr


def gaussian_logprob(miu, log_sigma, delta, z, eps: float = 1e-9) -> Tensor:
    """complex_standard_vae_loss.cal_gaussian_prob, model/pretrain_pvaes_loss.py:64-127: log q(z | miu, sigma, delta) of the
    improper complex Gaussian per (posterior b, sample s, frame t); miu / log_sigma / delta [B, T, H, 2], z [B or 1, S, T, H, 2]
    (broadcast over the posterior axis) -> [B, S, T].  Note the guard scales to 0.90 sigma here (0.99 in the KL)."""
    u = lambda v: v.unsqueeze(1)
    sg = u(torch.exp(log_sigma[..., 0]))
    dr, di = u(delta[..., 0]), u(delta[..., 1])
    a = torch.sqrt(dr * dr + di * di + eps)
    scale = sg * 0.90 / (a + eps)
    hit = a >= (sg - 1e-3)
    dr, di = torch.where(hit, dr * scale, dr), torch.where(hit, di * scale, di)
    q = dr * dr + di * di
    P = sg - q / (sg + eps)
    rp = 1 / (P + eps)
    Rr = dr / (sg * P + eps)
    Ri = -di / (sg * P + eps)
    m = rp - q / (sg * P * sg + eps)
    logs = torch.sum(torch.log(m + eps), dim=3) + torch.sum(torch.log(rp + eps), dim=3)
    xr, xi = z[..., 0] - u(miu[..., 0]), z[..., 1] - u(miu[..., 1])
    quad = torch.sum((xr * xr - xi * xi) * Rr - 2 * xr * xi * Ri, dim=3) - torch.sum((xr * xr + xi * xi) * rp, dim=3)
    return 0.5 * logs + quad


def mutual_information(miu, log_sigma, delta, z, num_samples: int, eps: float = 1e-9) -> Tensor:
    """complex_standard_vae_loss.mutual_information, model/pretrain_pvaes_loss.py:129-159: the minibatch estimate of I(x; z),
    mean over (i, s, t) of log q(z_ist | x_i) - (logsumexp_j log q(z_ist | x_j) - log B); z [B * num_samples, T, H, 2]."""
    B, T, H, D = miu.shape
    z = z.view(B, num_samples, T, H, D)
    log_q_zx = gaussian_logprob(miu, log_sigma, delta, z, eps)
    log_q_z = torch.stack([torch.logsumexp(gaussian_logprob(miu, log_sigma, delta, z[i].unsqueeze(0), eps), dim=0)
                           - torch.log(torch.tensor(float(B))) for i in range(B)])
    return torch.mean(log_q_zx - log_q_z)


def cvae_elbo_full(source, est, stft_source, pred_ri, miu, log_sigma, delta, z, kl_weight: float, mi_weight: float,
                   recon_loss_type: str, weights: Sequence[float], num_samples: int, prior_mode: str):
    """complex_standard_vae_loss.cal_loss with all its branches (model/pretrain_pvaes_loss.py:313-347): recon 'multiple' | 'prob'
    (:161-182), prior 'ri_inde' | 'ri_corr' (:322-331), minus mi_weight x the mutual-information estimate (:334-343)."""
    if recon_loss_type == "multiple":
        recon, l_cpx, l_mag, l_snr = multiple_recon_loss(pred_ri, stft_source, source, est, weights)
    else:
        d = (pred_ri[..., 0] - stft_source[..., 0]) ** 2 + (pred_ri[..., 1] - stft_source[..., 1]) ** 2
        recon = torch.sum(d, dim=1).mean()
        l_cpx = l_mag = l_snr = torch.tensor(0)
    zz = torch.zeros_like
    d_prior = zz(delta)
    if prior_mode == "ri_corr":
        d_prior[..., 1] = 1
    kl = complex_kl(miu, zz(miu), log_sigma, zz(log_sigma), delta, d_prior, 1e-9).mean()
    mi = mutual_information(miu, log_sigma, delta, z, num_samples) if mi_weight != 0 else torch.tensor(0)
    return recon + kl_weight * kl - mi_weight * mi, recon, kl, mi, l_cpx, l_mag, l_snr


def nsvae_loss(mc, mn, ms, mnn, lc, ln, ls, lnn, dc, dn, ds, dnn, alpha, w_kl, w_dismiu, latent_num=2):
    """standard_nsvae_loss_true_kl.final_nsvae_loss with w_resi=0
    (model/nsvae_loss.py:330-360, :448-473).  c=clean, n=noise, s=noisy-speech, nn=noisy-noise."""
    kl_c = complex_kl(ms, mc, ls, lc, ds, dc, 1e-10).mean()
    if latent_num == 2:
        kl_n = complex_kl(mnn, mn, lnn, ln, dnn, dn, 1e-10).mean()
        kl = kl_c + alpha * kl_n
    else:
        kl_n = complex_kl(ms, mn, ls, ln, ds, dn, 1e-10).mean()
        kl = kl_c - alpha * kl_n
    dis_s = torch.sqrt(torch.sum(torch.mean((mc - ms) ** 2, dim=(0, 1))))
    dis_n = torch.sqrt(torch.sum(torch.mean((mn - mnn) ** 2, dim=(0, 1))))
    return w_kl * kl + w_dismiu * (dis_s + dis_n), kl, kl_c, kl_n, dis_s, dis_n


# --------------------------------------------------------------------------- #
# mask application                                                            #
# --------------------------------------------------------------------------- #
def apply_mask(mask: Tensor, stft_ri: Tensor) -> Tensor:
    """model/pvae_module.py:224-234 (= :2594-2608): tanh-bounded magnitude mask plus
    phase rotation, in the reference's atan2/exp form.  mask and stft_ri are
    [B, F, T, 2]; returns predict [B, F, T, 2]."""
    mr, mi = mask[..., 0], mask[..., 1]
    mag = torch.tanh(torch.sqrt(mr * mr + mi * mi))
    ph = torch.atan2(mi / (mag + 1e-8), mr / (mag + 1e-8))
    xr, xi = stft_ri[..., 0], stft_ri[..., 1]
    xmag = torch.sqrt(xr * xr + xi * xi)
    xph = torch.atan2(xi, xr)
    amp = xmag * mag
    return torch.stack((amp * torch.cos(xph + ph), amp * torch.sin(xph + ph)), dim=-1)


# --------------------------------------------------------------------------- #
# network tables (model/net_config.py, model/causal_netconfig.py)             #
# --------------------------------------------------------------------------- #
# --------------------------------------------------------------------------- #
# Evaluation-path estimators.  The evaluation script itself is not importable here (librosa / soundfile /
# pesq at module level); tests/golden/make_golden.py `evalpath` extracts exactly these pure-torch / pure-numpy function
# definitions from the reference file's syntax tree, runs them, and pins the restatements below (op_outtype.npz, op_sisdr.npz).
def outtype_real_imag_mask(noise_c: Tensor, speech_c: Tensor, noisy_ri: Tensor) -> Tensor:
    """i_dccrn_vae/nsvae_dccrn/test_se_cvaefinetune.py:85-101.  noise_c / speech_c: complex [ns, F, T]; noisy_ri [1, F, T, 2]."""
    n = torch.view_as_real(noise_c).mean(dim=0)
    sp = torch.view_as_real(speech_c).mean(dim=0)
    x = noisy_ri.mean(dim=0)
    mr = sp[..., 0] ** 2 / (sp[..., 0] ** 2 + n[..., 0] ** 2 + 1e-10)
    mi = sp[..., 1] ** 2 / (sp[..., 1] ** 2 + n[..., 1] ** 2 + 1e-10)
    return torch.complex(mr * x[..., 0], mi * x[..., 1])


def outtype_complex_mask(noise_c: Tensor, speech_c: Tensor, noisy_ri: Tensor) -> Tensor:
    """test_se_cvaefinetune.py:104-116 (the eps is added to the complex denominator as a real number)."""
    x = torch.complex(noisy_ri[..., 0], noisy_ri[..., 1]).squeeze(0)
    n, sp = noise_c.mean(dim=0), speech_c.mean(dim=0)
    return sp / (sp + n + 1e-10) * x


def outtype_phase_sensitive_mask(noise_c: Tensor, speech_c: Tensor, noisy_ri: Tensor) -> Tensor:
    """test_se_cvaefinetune.py:119-135: |S| / (|S| + |N| + eps) * cos(angle S - angle X) * |X| * exp(j angle S)."""
    sp = speech_c.mean(dim=0)
    n = noise_c.mean(dim=0)
    x = torch.complex(noisy_ri[..., 0], noisy_ri[..., 1]).squeeze(0)
    sph, smag, nmag = torch.angle(sp), torch.abs(sp), torch.abs(n)
    mask = smag / (smag + nmag + 1e-10) * torch.cos(sph - torch.angle(x))
    return mask * torch.abs(x) * torch.exp(1j * sph)


def sisdr_np(x_est, x_ref):
    """utils/eval_metrics.py:49-64 (numpy, one estimate / one reference; eps = machine epsilon of the estimate's dtype)."""
    import numpy as np
    eps = np.finfo(x_est.dtype).eps
    ref = x_ref.reshape(x_ref.size, 1)
    est = x_est.reshape(x_est.size, 1)
    rss = np.dot(ref.T, ref)
    a = (eps + np.dot(ref.T, est)) / (rss + eps)
    e_true = a * ref
    e_res = est - e_true
    return 10 * np.log10((eps + (e_true ** 2).sum()) / (eps + (e_res ** 2).sum()))


def residual_loss(skiper_clean, skiper_noise, skiper_noisy, skip_to_use, latent_num, skiper_split, matching):
    """standard_nsvae_loss_true_kl.residual_loss, model/nsvae_loss.py:363-446: per used skip connection the mean squared
    difference between the clean (noise) encoder's skip and the noisy encoder's (its first / second half of the channels when
    split) -> (total, speech, noise).  Pinned by op_resi.npz (make_golden.py resi: the reference class itself)."""
    n = len(skiper_clean)
    split = skiper_split if (latent_num == 1 or matching == "speech") else True
    speech, noise = 0, 0
    for idx in range(n):
        if (n - 1 - idx) not in skip_to_use:
            continue
        y = skiper_noisy[idx]
        half = y.shape[1] // 2
        ys = y[:, :half] if split else y
        speech = speech + torch.mean((skiper_clean[idx] - ys).pow(2))
        if latent_num == 2 and matching == "both":
            noise = noise + torch.mean((skiper_noise[idx] - y[:, half:]).pow(2))
    return speech + noise, speech, noise


def net_params(causal: bool, base: int = 32, zdim_dense: int = 128) -> dict:
    """Shape table of model/causal_netconfig.py:5-103 / model/net_config.py:5-103
    (they differ only in the encoder time padding: 1 causal, 0 otherwise).  ``base``
    scales the channel widths for reduced-size fixtures (32 = the reference's)."""
    enc = [1, base, 2 * base, 4 * base, 4 * base, 8 * base, 8 * base]
    dec = [8 * base, 8 * base, 4 * base, 4 * base, 2 * base, base, 1]
    n = 6
    return {
        "encoder_channels": enc,
        "encoder_kernel_sizes": [(5, 2)] * n,
        "encoder_strides": [(2, 1)] * n,
        "encoder_paddings": [(2, 1 if causal else 0)] * n,
        "lstm_dim": [8 * base * 5, zdim_dense],
        "dense": [zdim_dense, 8 * base * 5],
        "lstm_layer_num": 2,
        "decoder_channels": dec,
        "decoder_kernel_sizes": [(5, 2)] * n,
        "decoder_strides": [(2, 1)] * n,
        "decoder_paddings": [(2, 0)] * n,
        "encoder_chw": [(enc[i + 1], 0, 0) for i in range(n)],
        "decoder_chw": [(dec[i + 1], 0, 0) for i in range(n)],
    }


# --------------------------------------------------------------------------- #
# model assembly (functional, driven by a reference-keyed state_dict)          #
# --------------------------------------------------------------------------- #
class BNState:
    """Collects the batch statistics a train-mode pass produces (the reference writes
    them into the running buffers, complex_progress.py:144-159)."""

    def __init__(self):
        self.stats: Dict[str, Tuple[Tensor, ...]] = {}


def _cbn(x, sd, pre, train, bn_state: Optional[BNState]):
    g = lambda k: sd[pre + k]
    if train:
        st = cbn_batch_stats(x)
        if bn_state is not None:
            bn_state.stats[pre] = st
    else:
        st = (g("running_mean_real"), g("running_mean_imag"), g("Vrr"), g("Vri"), g("Vii"))
    return cbn_whiten_affine(x, *st, g("gamma_rr"), g("gamma_ri"), g("gamma_ii"), g("beta_r"), g("beta_i"))


def encoder_block(x, sd, pre, np_, idx, causal, train, bn_state=None):
    """Encoder.forward, model/pvae_module.py:64-68."""
    y = complex_conv2d(x, sd[pre + "conv.conv_re.weight"], sd[pre + "conv.conv_re.bias"],
                       sd[pre + "conv.conv_im.weight"], sd[pre + "conv.conv_im.bias"],
                       np_["encoder_strides"][idx], np_["encoder_paddings"][idx], causal)
    y = _cbn(y, sd, pre + "bn.", train, bn_state)
    return prelu(y, sd[pre + "prelu.weight"])


def decoder_block(x, sd, pre, np_, idx, causal, train, bn_state=None):
    """Decoder.forward, model/pvae_module.py:88-93 (if_bn is always True in the
    shipped classes)."""
    y = complex_conv_transpose2d(x, sd[pre + "transconv.tconv_re.weight"], sd[pre + "transconv.tconv_re.bias"],
                                 sd[pre + "transconv.tconv_im.weight"], sd[pre + "transconv.tconv_im.bias"],
                                 np_["decoder_strides"][idx], np_["decoder_paddings"][idx], causal)
    y = _cbn(y, sd, pre + "bn.", train, bn_state)
    return prelu(y, sd[pre + "prelu.weight"])


def run_encoders(x5, sd, pre, np_, causal, train, bn_state=None):
    skips = []
    for i in range(len(np_["encoder_channels"]) - 1):
        x5 = encoder_block(x5, sd, f"{pre}encoders.{i}.", np_, i, causal, train, bn_state)
        skips.append(x5)
    return x5, skips


def standard_dccrn(x5, sd, pre, np_, causal, skip_to_use, train, bn_state=None):
    """standard_DCCRN.forward, model/pvae_module.py:174-198.  Returns (mask, latent)."""
    x, skips = run_encoders(x5, sd, pre, np_, causal, train, bn_state)
    B, C, F, T, D = x.shape
    l = x.reshape(B, C * F, T, D).permute(2, 0, 1, 3)
    l = complex_lstm(l, sd, pre + "lstms.0.", np_["lstm_layer_num"])
    l = l.permute(1, 0, 2, 3)
    latent = l
    d = complex_dense(l.reshape(B * T, -1, D), sd[pre + "dense.linear_read.weight"], sd[pre + "dense.linear_read.bias"],
                      sd[pre + "dense.linear_imag.weight"], sd[pre + "dense.linear_imag.bias"])
    p = d.reshape(B, T, C, F, D).permute(0, 2, 3, 1, 4)
    for i in range(len(np_["decoder_channels"]) - 1):
        if i in skip_to_use:
            p = torch.cat([p, skips[len(skips) - 1 - i]], dim=1)
        p = decoder_block(p, sd, f"{pre}decoders.{i}.", np_, i, causal, train, bn_state)
    return p, latent


def dccrn_forward(signal, sd, np_, causal, n_fft, hop, win_length, skip_to_use, recon_type="mask",
                  train=False, bn_state=None, data_mean=None, data_std=None):
    """DCCRN_.forward, model/pvae_module.py:215-255 (no resynthesis).  data_mean / data_std [1, F, 1, 2]: the optional input
    normalisation (:217-221: (stft - mean) / (std + 1e-6), imaginary parts of the first and last bin zeroed) and its inverse on
    the prediction (:235-238, :246-247).  Returns (clean [B, L'], predict [B,F,T,2], latent)."""
    X = stft(signal, n_fft, hop, win_length)
    if data_mean is not None:
        X = (X - data_mean) / (data_std + 1e-6)
        X = X.clone()
        X[:, 0, :, 1] = 0
        X[:, -1, :, 1] = 0
    out, latent = standard_dccrn(X.unsqueeze(1), sd, "std_DCCRN.", np_, causal, skip_to_use, train, bn_state)
    out = out.squeeze(1)
    pred = apply_mask(out, X) if recon_type == "mask" else out
    if data_mean is not None:
        pred = data_std * pred + data_mean
    return istft(pred, n_fft, hop, win_length), pred, latent


def vae_encoder_forward(signal, sd, np_, causal, zdim, n_fft, hop, win_length, num_samples, latent_num,
                        eps: Sequence[Tensor], train=False, bn_state=None):
    """pvae_dccrn_encoder_skip_prepare.forward (pvae_module.py:1888-1914, latent_num=1,
    8-tuple) and nsvae_pvae_dccrn_encoder_twophase.forward (:2233-2268, 12-tuple).
    ``eps`` = [eps_r, eps_i] per latent (injected randn draws).  Returns a dict."""
    X = stft(signal, n_fft, hop, win_length)
    x, skips = run_encoders(X.unsqueeze(1), sd, "", np_, causal, train, bn_state)
    B, C, F, T, D = x.shape
    l = x.reshape(B, C * F, T, D).permute(2, 0, 1, 3)
    l = complex_lstm(l, sd, "lstms.0.", np_["lstm_layer_num"]).permute(1, 0, 2, 3)
    out = {"skiper": skips, "C": C, "F": F, "stft_x": X, "lstm_out": l}
    names = ["speech", "noise"][:latent_num]
    for k, nm in enumerate(names):
        o = 3 * zdim * k
        miu, ls, dl = l[:, :, o:o + zdim], l[:, :, o + zdim:o + 2 * zdim], l[:, :, o + 2 * zdim:o + 3 * zdim]
        out[f"miu_{nm}"], out[f"log_sigma_{nm}"], out[f"delta_{nm}"] = miu, ls, dl
        out[f"z_{nm}"] = reparameterization(miu, ls, dl, num_samples, eps[2 * k], eps[2 * k + 1])
    return out


def vae_decoder_forward(stft_x, z, skiper, C, F, sd, np_, causal, num_samples, n_fft, hop, win_length,
                        recon_type, skip_to_use, pad="zero", use_sc=True, train=False, bn_state=None):
    """pvae_dccrn_decoder_skip_prepare.forward (pvae_module.py:2082-2122: always zero
    skips) and nsvae_pvae_dccrn_decoder_twophase.forward (:2548-2619: pad='zero'|'sig',
    recon_type 'real_imag'|'mask').  Returns (recon_sig, predict [B*ns,F,T,2])."""
    Bn, T, zdim, D = z.shape
    d = complex_dense(z.reshape(Bn * T, zdim, D), sd["dense.linear_read.weight"], sd["dense.linear_read.bias"],
                      sd["dense.linear_imag.weight"], sd["dense.linear_imag.bias"])
    p = d.reshape(Bn, T, C, F, D).permute(0, 2, 3, 1, 4)
    for i in range(len(np_["decoder_channels"]) - 1):
        if use_sc and i in skip_to_use:
            sk = skiper[len(skiper) - 1 - i]
            if pad == "zero":
                sk = torch.zeros((Bn,) + tuple(sk.shape[1:]), dtype=p.dtype)
            else:
                sk = sk.repeat_interleave(num_samples, dim=0)
            p = torch.cat([p, sk], dim=1)
        p = decoder_block(p, sd, f"decoders.{i}.", np_, i, causal, train, bn_state)
    out = p.squeeze(1)
    if recon_type == "mask":
        pred = apply_mask(out, stft_x.repeat_interleave(num_samples, dim=0))
    else:
        pred = out
    return istft(pred, n_fft, hop, win_length), pred


# --------------------------------------------------------------------------- #
# deterministic synthetic weights: one generator shared with the product-side      #
# utilities (i-dccrn-vae_amd/utils/synth.py) so fixtures, tests and bench agree    #
# --------------------------------------------------------------------------- #
def _synth():
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    return importlib.import_module("i-dccrn-vae_amd.utils.synth")


def synth_tensor(name, shape, seed):
    return _synth().synth_tensor(name, shape, seed)


def synth_state_dict(shapes, seed):
    return _synth().synth_state_dict(shapes, seed)
