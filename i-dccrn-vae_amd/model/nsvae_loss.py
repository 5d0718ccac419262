"""NSVAE / supervised / two-phase losses (reference: model/nsvae_loss.py: standard_nsvae_loss_true_kl :243-473,
ete_train_se_loss :755-806, two_phase_loss :809-948) on the HIP reduction kernels."""
import torch

from .. import autograd as AG
from .. import ops
from ._loss_common import kl_mean, latent_ref, recon_terms
from .complex_progress import planar_of
from .sisnr_loss import si_snr as _si_snr


def _miu_dist(miu_a, miu_b):
    pa, oa = latent_ref(miu_a, miu_a, miu_a) if getattr(miu_a, "_idv", None) is None else (miu_a._idv, (miu_a._idv_off,) * 3)
    pb, ob = latent_ref(miu_b, miu_b, miu_b) if getattr(miu_b, "_idv", None) is None else (miu_b._idv, (miu_b._idv_off,) * 3)
    if AG.grad_mode(pa.buf, pb.buf):
        return AG.MiuDistFn.apply(AG._geom(pa), oa[0], AG._geom(pb), ob[0], miu_a.shape[2], pa.buf, pb.buf)
    return ops.miu_dist(pa, oa[0], pb, ob[0], miu_a.shape[2])


def _msd(a, ca0, b, cb0, C):
    """mean over [B, C, F, T, 2] of (a[:, ca0:ca0+C] - b[:, cb0:cb0+C])^2 for two planar activations (idv_msd)."""
    if (a.F, a.B, a.T, a.Tp) != (b.F, b.B, b.T, b.Tp):
        raise RuntimeError(f"residual loss: skip connections of different shape ({(a.B, a.F, a.T)} vs {(b.B, b.F, b.T)})")
    return ops.msd(a, ca0, b, cb0, C)


class _KLMixin:
    def cal_kl(self, miu1, miu2, log_sigma1, log_sigma2, delta1, delta2, z1):
        """Closed-form KL(q1 || q2) (reference :275-328); z1 is accepted and ignored, as there.
        Returns the mean over (b, t) - every caller in the reference takes torch.mean of the [B, T] map."""
        return kl_mean((miu1, log_sigma1, delta1), (miu2, log_sigma2, delta2), miu1.shape[2], self.epsilon)

    def _kl_pair(self, c, n, s, nn, alpha, sign_latent1):
        kl_clean = self.cal_kl(s[0], c[0], s[1], c[1], s[2], c[2], None)
        if self.latent_num == 1:
            kl_noise = self.cal_kl(s[0], n[0], s[1], n[1], s[2], n[2], None)
            return kl_clean + sign_latent1 * alpha * kl_noise, kl_clean, kl_noise
        kl_noise = self.cal_kl(nn[0], n[0], nn[1], n[1], nn[2], n[2], None)
        return kl_clean + alpha * kl_noise, kl_clean, kl_noise


class standard_nsvae_loss_true_kl(_KLMixin):
    def __init__(self, alpha, w_resi, w_kl, w_dismiu, zdim, num_samples, latent_num, nsvae_model, skipc, skip_to_use,
                 matching):
        self.alpha, self.w_resi, self.w_kl, self.w_dismiu = alpha, w_resi, w_kl, w_dismiu
        self.epsilon = 1e-10
        self.num_samples, self.latent_num = num_samples, latent_num
        self.skip_to_use, self.nsvae_model, self.skipc = skip_to_use, nsvae_model, skipc
        self.skiper_split = nsvae_model in ('adapt', 'double')
        self.matching = matching
        self.zdim = zdim

    def kl_loss(self, miu_clean, miu_noise, miu_noisy_speech, miu_noisy_noise, log_sigma_clean, log_sigma_noise,
                log_sigma_noisy_speech, log_sigma_noisy_noise, delta_clean, delta_noise, delta_noisy_speech,
                delta_noisy_noise, z_noisy_speech, z_noisy_noise):
        """reference :330-347"""
        return self._kl_pair((miu_clean, log_sigma_clean, delta_clean), (miu_noise, log_sigma_noise, delta_noise),
                             (miu_noisy_speech, log_sigma_noisy_speech, delta_noisy_speech),
                             (miu_noisy_noise, log_sigma_noisy_noise, delta_noisy_noise), self.alpha, -1.0)

    def miu_dis_loss(self, miu_clean, miu_noise, miu_noisy_speech, miu_noisy_noise, *unused):
        """reference :349-360"""
        speech = _miu_dist(miu_clean, miu_noisy_speech)
        noise = _miu_dist(miu_noise, miu_noisy_noise)
        return speech + noise, speech, noise

    def final_nsvae_loss(self, miu_clean, miu_noise, miu_noisy_speech, miu_noisy_noise, log_sigma_clean, log_sigma_noise,
                         log_sigma_noisy_speech, log_sigma_noisy_noise, delta_clean, delta_noise, delta_noisy_speech,
                         delta_noisy_noise, z_noisy_speech, z_noisy_noise, skiper_clean, skiper_noise, skiper_noisy):
        """reference :448-473 -> (final, kl, kl_clean, kl_noise, dismiu_speech, dismiu_noise, resi, resi_speech, resi_noise)"""
        kl_loss, kl_clean, kl_noise = self.kl_loss(miu_clean, miu_noise, miu_noisy_speech, miu_noisy_noise,
                                                   log_sigma_clean, log_sigma_noise, log_sigma_noisy_speech,
                                                   log_sigma_noisy_noise, delta_clean, delta_noise, delta_noisy_speech,
                                                   delta_noisy_noise, z_noisy_speech, z_noisy_noise)
        dismiu_loss, dismiu_speech, dismiu_noise = self.miu_dis_loss(miu_clean, miu_noise, miu_noisy_speech, miu_noisy_noise)
        final_loss = self.w_kl * kl_loss + self.w_dismiu * dismiu_loss
        resi = (0, 0, 0)
        if self.skipc == 'True' and self.w_resi != 0:
            # computed and returned, NOT added to the loss that is back-propagated: exactly the reference (:460-466)
            resi = self.residual_loss(skiper_clean, skiper_noise, skiper_noisy)
        return (final_loss, kl_loss, kl_clean, kl_noise, dismiu_speech, dismiu_noise) + tuple(resi)

    def residual_loss(self, skiper_clean, skiper_noise, skiper_noisy):
        """Skip-matching term, reference :363-446: per used skip connection the mean squared difference between the clean (and
        noise) encoder's skip and the noisy encoder's -- its first / second half of the channels where the model carries both
        (`skiper_split`, matching 'both') -> (total, speech, noise).  A value for the log, no gradient (see final_nsvae_loss)."""
        n = len(skiper_clean)
        used = [idx for idx in range(n) if (n - 1 - idx) in self.skip_to_use]
        split = self.skiper_split if (self.latent_num == 1 or self.matching == 'speech') else True
        both = self.latent_num == 2 and self.matching == 'both'
        if self.latent_num == 2 and self.matching not in ('both', 'speech'):
            raise ValueError(f"matching {self.matching!r}")
        speech = noise = 0
        for idx in used:
            c = planar_of(skiper_clean[idx])
            y = planar_of(skiper_noisy[idx], c.Tp)
            half = y.C // 2 if split else y.C
            if c.C != half:              # torch's own shape error for (connct - connct2) in the reference
                raise RuntimeError(f"residual loss: clean skip has {c.C} channels, the noisy encoder's {'half' if split else 'skip'} {half}")
            speech = speech + _msd(c, 0, y, 0, half)
            if both:
                nz = planar_of(skiper_noise[idx], c.Tp)
                if nz.C != y.C - half:
                    raise RuntimeError(f"residual loss: noise skip has {nz.C} channels, the noisy encoder's second half {y.C - half}")
                noise = noise + _msd(nz, 0, y, half, y.C - half)
        return speech + noise, speech, noise


class ete_train_se_loss():
    """Supervised DCCRN loss, reference :755-806."""

    def __init__(self, recon_loss_weight):
        self.recon_loss_weight = recon_loss_weight
        self.epsilon = 1e-10

    def si_snr(self, source, estimate_source, eps=1e-8):
        return _si_snr(source, estimate_source, eps)

    def multiple_recon_loss(self, predict_cpx_stft, ori_cpx_stft, source, est_source):
        return recon_terms(predict_cpx_stft, ori_cpx_stft, source, est_source, self.recon_loss_weight)

    def final_ete_loss(self, predict_cpx_stft, ori_cpx_stft, source, est_source):
        return self.multiple_recon_loss(predict_cpx_stft, ori_cpx_stft, source, est_source)


class two_phase_loss(_KLMixin):
    """reference :809-948"""

    def __init__(self, recon_loss_weight, alpha, zdim, latent_num):
        self.epsilon = 1e-10
        self.recon_loss_weight = recon_loss_weight
        self.alpha, self.zdim, self.latent_num = alpha, zdim, latent_num

    def si_snr(self, source, estimate_source, eps=1e-8):
        return _si_snr(source, estimate_source, eps)

    def multi_recon_loss(self, predict_cpx_stft, ori_cpx_stft, source, est_source):
        return recon_terms(predict_cpx_stft, ori_cpx_stft, source, est_source, self.recon_loss_weight)

    def phase_2_loss(self, predict_stft_clean, stft_x_clean, clean_batch, recon_sig_clean, predict_stft_noise,
                     stft_x_noise, noise_batch, recon_sig_noise):
        """reference :916-927"""
        fc, cpx_c, mag_c, snr_c = self.multi_recon_loss(predict_stft_clean, stft_x_clean, clean_batch, recon_sig_clean)
        if self.latent_num == 1:
            zero = torch.tensor([0])
            return fc, cpx_c, mag_c, snr_c, zero, zero, zero
        fn, cpx_n, mag_n, snr_n = self.multi_recon_loss(predict_stft_noise, stft_x_noise, noise_batch, recon_sig_noise)
        return fc + fn, cpx_c, mag_c, snr_c, cpx_n, mag_n, snr_n

    def phase_1_loss(self, miu_clean, miu_noise, miu_noisy_speech, miu_noisy_noise, log_sigma_clean, log_sigma_noise,
                     log_sigma_noisy_speech, log_sigma_noisy_noise, delta_clean, delta_noise, delta_noisy_speech,
                     delta_noisy_noise, z_noisy_speech, z_noisy_noise):
        """reference :931-948 (latent_num 2 adds the noise KL with weight 1, latent_num 1 subtracts alpha * KL)"""
        return self._kl_pair((miu_clean, log_sigma_clean, delta_clean), (miu_noise, log_sigma_noise, delta_noise),
                             (miu_noisy_speech, log_sigma_noisy_speech, delta_noisy_speech),
                             (miu_noisy_noise, log_sigma_noisy_noise, delta_noisy_noise),
                             1.0 if self.latent_num == 2 else self.alpha, -1.0)
