"""Shared helpers of the loss classes: resolve reference-layout tensors to the planar buffers the HIP
reduction kernels read, and the three reconstruction terms."""
from __future__ import annotations

import torch

from .. import autograd as AG
from .. import ops
from ..ops import Planar
from .sisnr_loss import si_snr


def latent_ref(miu, log_sigma, delta, Tp=None):
    """(planar latent, (off_miu, off_log_sigma, off_delta)) for three [B, T, H, 2] tensors.  Views handed out
    by the encoders share one planar LSTM output and cost nothing; foreign tensors are packed once."""
    pls = [getattr(t, "_idv", None) for t in (miu, log_sigma, delta)]
    if pls[0] is not None and pls[0] is pls[1] and pls[0] is pls[2]:
        return pls[0], (miu._idv_off, log_sigma._idv_off, delta._idv_off)
    if not miu.is_cuda:
        raise RuntimeError("i-dccrn-vae_amd runs on the MI355X only: pass CUDA (ROCm) tensors")
    H = miu.shape[2]
    lat = torch.cat([miu, log_sigma, delta], dim=2).permute(0, 2, 1, 3).unsqueeze(2)
    return Planar.from_tensor5(lat.float(), Tp), (0, H, 2 * H)


def kl_mean(q1, q2, zdim: int, eps: float) -> torch.Tensor:
    """mean over (b, t) of KL(q1 || q2); q = (miu, log_sigma, delta); q2 None = prior (0, 0, 0)."""
    p1, o1 = latent_ref(*q1)
    p2, o2 = (None, None) if q2 is None else latent_ref(*q2, Tp=p1.Tp)
    if p2 is not None and (p1.B, p1.T, p1.Tp) != (p2.B, p2.T, p2.Tp):
        raise RuntimeError("KL operands must share batch and frame counts")
    if AG.grad_mode(p1.buf, p2.buf if p2 is not None else None):
        return AG.CklFn.apply(AG._geom(p1), tuple(o1), AG._geom(p2) if p2 is not None else None,
                              tuple(o2) if o2 is not None else None, zdim, eps, p1.buf, p2.buf if p2 is not None else None)
    return ops.ckl(p1, o1, p2, o2, zdim, eps)


def recon_terms(predict_cpx_stft, ori_cpx_stft, source, est_source, weights, with_sisnr: bool = True):
    """multiple_recon_loss of the reference (nsvae_loss.py:775-797 and copies): (final, cpx, mag, sisnr)."""
    pc = predict_cpx_stft
    if not torch.view_as_real(pc).is_contiguous():
        pc = pc.contiguous()
    ops.check_dev_f32(ori_cpx_stft, "ori_cpx_stft", pc.device)
    div = 1
    if ori_cpx_stft.shape[0] != pc.shape[0]:
        div = pc.shape[0] // ori_cpx_stft.shape[0]
    if AG.grad_mode(pc):
        loss_cpx, loss_mag = AG.ReconLossFn.apply(torch.view_as_real(pc), ori_cpx_stft.detach().float(), div)
    else:
        loss_cpx, loss_mag = ops.recon_loss(pc, ori_cpx_stft.float(), div)
    if not with_sisnr:
        return weights[0] * loss_cpx + weights[1] * loss_mag, loss_cpx, loss_mag, None
    sisnr = si_snr(source, est_source)
    final = weights[0] * loss_cpx + weights[1] * loss_mag + weights[2] * sisnr
    return final, loss_cpx, loss_mag, sisnr
