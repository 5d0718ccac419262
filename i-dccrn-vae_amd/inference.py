"""Enhancement inference as the reference's evaluation scripts run it, batched on the GPU.

  * supervised (supervised_dccrn/test.py:123-137): ``model(noisy, train=False)[0]``;
  * I-DCCRN-VAE (i_dccrn_vae/nsvae_dccrn/test_se_cvaefinetune.py:251-311): noisy encoder (eval) -> fine-tuned decoder with
    the noisy skips (``pad='sig'``) on ``num_samples`` (10 in test_se_cvaefinetune.sh) latent draws -> mean over the sampled
    waveforms.  The reference feeds one utterance at a time (``tmp_x[None]``); here a batch of equal-length utterances
    goes through at once (utterances are independent in eval mode: folded batch norm);
  * ``compute_sisdr`` (utils/eval_metrics.py:49-64) on the device.  PESQ / ESTOI / DNSMOS are third-party CPU metrics and
    stay out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from ._lib import call, p, i, stream_ptr


def mean_over_samples(recon: torch.Tensor, num_samples: int) -> torch.Tensor:
    """[B*ns, L] -> [B, L]: ``torch.mean(recon_sig_clean, dim=0)`` per utterance (test_se_cvaefinetune.py:309-311)."""
    ops.check_dev_f32(recon, "recon")
    Bn, L = recon.shape
    if Bn % num_samples:
        raise ValueError("batch is not a multiple of num_samples")
    recon = recon.float().contiguous()
    out = torch.empty(Bn // num_samples, L, dtype=torch.float32, device=recon.device)
    call("idv_mean_over_samples", p(recon), i(num_samples), i(Bn // num_samples), i(L), p(out), stream_ptr())
    return out


@torch.no_grad()
def enhance_supervised(model, noisy: torch.Tensor) -> torch.Tensor:
    """DCCRN / DCCRN-CL: [B, L] -> enhanced [B, hop*(T-1)]."""
    return model(noisy, train=False)[0]


@torch.no_grad()
def enhance_vae(noisy_encoder, decoder, noisy: torch.Tensor, eps=None, latent: str = "speech") -> torch.Tensor:
    """I-DCCRN-VAE (phase 2, latent_to_use 1): [B, L] -> mean of the num_samples decoded waveforms, [B, hop*(T-1)].
    ``eps``: optional injected Gaussian draws (see the encoder's forward)."""
    r = noisy_encoder(noisy, train=False, eps=eps)
    z = r[0] if latent == "speech" else r[4]
    if z is None:
        raise ValueError("this encoder has no noise latent (latent_num == 1)")
    skiper, C, F, stft_x = r[8], r[9], r[10], r[11]
    recon, _ = decoder(stft_x, z, skiper, C, F, train=False, pad="sig")
    return mean_over_samples(recon, noisy_encoder.num_samples)


def compute_sisdr(x_est: torch.Tensor, x_ref: torch.Tensor) -> torch.Tensor:
    """SI-SDR in dB per utterance (utils/eval_metrics.py:49-64); inputs [L] or [B, L] on the GPU -> tensor [B] (or scalar)."""
    ops.check_dev_f32(x_est, "x_est")
    ops.check_dev_f32(x_ref, "x_ref", x_est.device)
    single = x_est.dim() == 1
    e = x_est.reshape(1, -1) if single else x_est
    r = x_ref.reshape(1, -1) if single else x_ref
    if e.shape != r.shape:
        raise ValueError(f"estimate {tuple(e.shape)} and reference {tuple(r.shape)} differ")
    e, r = e.float().contiguous(), r.float().contiguous()
    B, L = e.shape
    work = torch.empty(3 * B, dtype=torch.float64, device=e.device)
    out = torch.empty(B, dtype=torch.float32, device=e.device)
    call("idv_sisdr", p(r), i(r.stride(0)), p(e), i(e.stride(0)), i(B), i(L), p(work), p(out), stream_ptr())
    return out[0] if single else out
