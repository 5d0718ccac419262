"""NSVAE / supervised / two-phase losses (reference: model/nsvae_loss.py: standard_nsvae_loss_true_kl :243-473,
ete_train_se_loss :755-806, two_phase_loss :809-948) on the HIP reduction kernels."""
import torch

from .. import autograd as AG
from .. import ops
from ._loss_common import kl_mean, latent_ref, recon_terms
from .sisnr_loss import si_snr as _si_snr


def _miu_dist(miu_a, miu_b):
    pa, oa = latent_ref(miu_a, miu_a, miu_a) if getattr(miu_a, "_idv", None) is None else (miu_a._idv, (miu_a._idv_off,) * 3)
    pb, ob = latent_ref(miu_b, miu_b, miu_b) if getattr(miu_b, "_idv", None) is None else (miu_b._idv, (miu_b._idv_off,) * 3)
    if AG.grad_mode(pa.buf, pb.buf):
        return AG.MiuDistFn.apply(AG._geom(pa), oa[0], AG._geom(pb), ob[0], miu_a.shape[2], pa.buf, pb.buf)
    return ops.miu_dist(pa, oa[0], pb, ob[0], miu_a.shape[2])


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
        if self.skipc == 'True' and self.w_resi != 0:
            raise NotImplementedError("residual (skip-matching) loss: w_resi is 0 in the shipped recipe")
        kl_loss, kl_clean, kl_noise = self.kl_loss(miu_clean, miu_noise, miu_noisy_speech, miu_noisy_noise,
                                                   log_sigma_clean, log_sigma_noise, log_sigma_noisy_speech,
                                                   log_sigma_noisy_noise, delta_clean, delta_noise, delta_noisy_speech,
                                                   delta_noisy_noise, z_noisy_speech, z_noisy_noise)
        dismiu_loss, dismiu_speech, dismiu_noise = self.miu_dis_loss(miu_clean, miu_noise, miu_noisy_speech, miu_noisy_noise)
        final_loss = self.w_kl * kl_loss + self.w_dismiu * dismiu_loss
        return final_loss, kl_loss, kl_clean, kl_noise, dismiu_speech, dismiu_noise, 0, 0, 0


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
