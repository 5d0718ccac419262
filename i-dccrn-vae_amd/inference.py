"""Enhancement inference as the reference's evaluation scripts run it, batched on the GPU.

  * supervised (supervised_dccrn/test.py:123-137): ``model(noisy, train=False)[0]``;
  * I-DCCRN-VAE (i_dccrn_vae/nsvae_dccrn/test_se_cvaefinetune.py:251-311): noisy encoder (eval) -> fine-tuned decoder with
    the noisy skips (``pad='sig'``) on ``num_samples`` (10 in test_se_cvaefinetune.sh) latent draws -> mean over the sampled
    waveforms.  The reference feeds one utterance at a time (``tmp_x[None]``); here a batch of equal-length utterances
    goes through at once (utterances are independent in eval mode: folded batch norm);
  * the two-latent evaluation (test_se_cvaefinetune.py:261-305, ``latent_to_use == 2``): speech AND noise decoders on their
    latents, then one of the ``outtype`` estimators -- ``clean_direct`` (mean of the sampled speech waveforms),
    ``real_imag_mask`` (:85-101), ``complex_mask`` (:104-116), ``phase_mask`` (:119-135) -- as one HIP kernel
    (``idv_outtype_estimate``) + the ISTFT (``torch.istft`` with the analysis window, :288 etc.);
  * ``compute_sisdr`` (utils/eval_metrics.py:49-64) on the device.  PESQ / ESTOI / DNSMOS are third-party CPU metrics and
    stay out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from ._lib import call, p, i, ll, stream_ptr


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
def enhance_supervised(model, noisy: torch.Tensor, check: bool = True) -> torch.Tensor:
    """DCCRN / DCCRN-CL: [B, L] -> enhanced [B, hop*(T-1)].  ``check`` (all three entry points): synchronise and raise
    ``ops.CoopTimeout`` here, at the operation that owns it, if a cooperative LSTM recurrence of this forward timed out
    (its outputs would be NaN); ``check=False`` keeps the call asynchronous (call ``ops.coop_check()`` yourself)."""
    return _checked(model(noisy, train=False)[0], check)


def _checked(out: torch.Tensor, check: bool) -> torch.Tensor:
    if check:
        ops.coop_check()
    return out


@torch.no_grad()
def enhance_vae(noisy_encoder, decoder, noisy: torch.Tensor, eps=None, latent: str = "speech", check: bool = True) -> torch.Tensor:
    """I-DCCRN-VAE (phase 2, latent_to_use 1): [B, L] -> mean of the num_samples decoded waveforms, [B, hop*(T-1)].
    ``eps``: optional injected Gaussian draws (see the encoder's forward)."""
    r = noisy_encoder(noisy, train=False, eps=eps)
    z = r[0] if latent == "speech" else r[4]
    if z is None:
        raise ValueError("this encoder has no noise latent (latent_num == 1)")
    skiper, C, F, stft_x = r[8], r[9], r[10], r[11]
    recon, _ = decoder(stft_x, z, skiper, C, F, train=False, pad="sig")
    return _checked(mean_over_samples(recon, noisy_encoder.num_samples), check)


OUTTYPES = {"real_imag_mask": 0, "complex_mask": 1, "phase_mask": 2}


def outtype_estimate(predict_noise: torch.Tensor, predict_speech: torch.Tensor, stft_noisy: torch.Tensor, outtype: str,
                     num_samples: int):
    """The mask estimators of test_se_cvaefinetune.py:85-135 for a batch: predict_* complex [B*ns, F, T] (the decoders'
    second output), stft_noisy [B, F, T, 2] (any strides) -> (planar spectrum for ops.istft, complex [B, F, T])."""
    if outtype not in OUTTYPES:
        raise ValueError(f"outtype {outtype!r}: expected one of {sorted(OUTTYPES)} (or 'clean_direct')")
    for t, n in ((predict_noise, "predict_noise"), (predict_speech, "predict_speech")):
        if not (t.is_cuda and t.dtype == torch.complex64):
            raise RuntimeError(f"{n} must be a complex64 tensor on the GPU (there is no CPU fallback)")
    ops.check_dev_f32(stft_noisy, "stft_noisy", predict_speech.device)
    Bn, F, T = predict_speech.shape
    if Bn % num_samples or predict_noise.shape != predict_speech.shape:
        raise ValueError("predict_speech / predict_noise must both be [B * num_samples, F, T]")
    B = Bn // num_samples
    if tuple(stft_noisy.shape) != (B, F, T, 2):
        raise ValueError(f"stft_noisy {tuple(stft_noisy.shape)}: expected {(B, F, T, 2)}")
    sp = torch.view_as_real(predict_speech.contiguous())
    no = torch.view_as_real(predict_noise.contiguous())
    out = ops.Planar.empty(1, F, B, T, T + 1, sp.device)
    oc = torch.empty(B, F, T, 2, dtype=torch.float32, device=sp.device)
    sb, sf, st_, sr = stft_noisy.stride()
    call("idv_outtype_estimate", p(sp), p(no), p(stft_noisy), ll(sb), ll(sf), ll(st_), ll(sr), i(OUTTYPES[outtype]), i(num_samples),
         i(B), i(F), i(T), i(out.Tp), i(out.Jp), out.ptr(), p(oc), stream_ptr())
    return out, torch.view_as_complex(oc)


@torch.no_grad()
def enhance_vae_two_latents(noisy_encoder, speech_decoder, noise_decoder, noisy: torch.Tensor, outtype: str = "clean_direct",
                            phase: int = 2, eps=None, check: bool = True) -> torch.Tensor:
    """latent_to_use == 2 (test_se_cvaefinetune.py:261-305): the noisy encoder's speech latent through the speech decoder and
    its noise latent through the noise decoder (phase 1: the pre-trained decoders, zero skips, :263-264; phase 2: the
    fine-tuned decoders with the noisy skips, ``pad='sig'``, :295-296), then the ``outtype`` estimator -> enhanced [B, L]."""
    r = noisy_encoder(noisy, train=False, eps=eps)
    if r[4] is None:
        raise ValueError("this encoder has no noise latent (latent_num == 1)")
    z_s, z_n, skiper, C, F, stft_x = r[0], r[4], r[8], r[9], r[10], r[11]
    kw = {"pad": "sig"} if phase == 2 else {}
    rec_s, pred_s = speech_decoder(stft_x, z_s, skiper, C, F, train=False, **kw)
    ns = noisy_encoder.num_samples
    if outtype == "clean_direct":
        return _checked(mean_over_samples(rec_s, ns), check)
    _, pred_n = noise_decoder(stft_x, z_n, skiper, C, F, train=False, **kw)
    spec, _ = outtype_estimate(pred_n, pred_s, stft_x, outtype, ns)
    from .model.pvae_module import dft_plan
    st = noisy_encoder.stft
    return _checked(ops.istft(spec, dft_plan(st.n_fft, st.win_length, st.hop_length, spec.T, spec.buf.device)), check)


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
