"""CVAE / NVAE pre-training ELBO (reference: model/pretrain_pvaes_loss.py: KL_annealing :3-42,
complex_standard_vae_loss :48-347 - the configuration the shipped recipe uses: recon_loss_type='multiple',
prior_mode 'ri_inde', mi_weight 0).  Reductions run in the idv_recon_loss / idv_sisnr / idv_ckl kernels."""
import torch

from ._loss_common import kl_mean, recon_terms
from .sisnr_loss import si_snr as _si_snr


class KL_annealing():
    """Cyclical linear KL warm-up weights (Fu et al. 2019), reference :3-42."""

    def __init__(self, kl_warm_epochs):
        self.kl_warm_epochs = kl_warm_epochs

    def frange_cycle_linear(self, start=0.0, stop=1.0, n_cycle=1, ratio=1):
        n = self.kl_warm_epochs
        sched = torch.ones(n) * stop
        period = n / n_cycle
        step = (stop - start) / (period * ratio)
        for c in range(n_cycle):
            v, k = start, 0
            while v <= stop and int(k + c * period) < n:
                sched[int(k + c * period)] = v
                v += step
                k += 1
        return sched


class complex_standard_vae_loss():
    def __init__(self, kl_warm_weights, kl_weight, mi_weight, recon_loss_type='prob', recon_type='real_imag',
                 recon_loss_weight=[1.0, 1.0, 1.0], num_samples=5, prior_mode='ri_inde'):
        self.kl_warm_weights = kl_warm_weights
        self.kl_warm_epochs = kl_warm_weights.size()[0]
        self.kl_weight = kl_weight
        self.epsilon = 1e-9
        self.recon_loss_type = recon_loss_type
        self.predict_type = recon_type
        self.recon_loss_weight = recon_loss_weight
        self.const = 1.14473
        self.num_samples = num_samples
        self.mi_weight = mi_weight
        self.prior_mode = prior_mode

    def si_snr(self, source, estimate_source, eps=1e-8):
        return _si_snr(source, estimate_source, eps)

    def multiple_recon_loss(self, predict_cpx_stft, ori_cpx_stft, source, est_source):
        return recon_terms(predict_cpx_stft, ori_cpx_stft, source, est_source, self.recon_loss_weight)

    def cal_kl_arbi_prior(self, miu1, miu2, log_sigma1, log_sigma2, delta1, delta2):
        """mean_{b,t} KL(q1 || q2), reference :225-281 (eps 1e-9)."""
        self.zdim = miu1.shape[2]
        return kl_mean((miu1, log_sigma1, delta1), (miu2, log_sigma2, delta2), self.zdim, self.epsilon)

    def cal_loss(self, source, est_source, stft_source, miu_x, miu, log_sigma, delta, z, epoch):
        """reference :313-347 -> (final, recon, kl, mi, loss_cpx, loss_mag, sisnr)"""
        if self.recon_loss_type != 'multiple':
            raise NotImplementedError("recon_loss_type 'multiple' is the one the shipped recipe uses")
        if self.prior_mode != 'ri_inde':
            raise NotImplementedError("prior_mode 'ri_inde' (standard prior) only")
        if self.mi_weight != 0:
            raise NotImplementedError("mi_weight != 0 (mutual-information term) is not on the shipped path")
        recon_loss, loss_cpx, loss_mag, sisnr = self.multiple_recon_loss(miu_x, stft_source, source, est_source)
        self.zdim = miu.shape[2]
        kl_loss = kl_mean((miu, log_sigma, delta), None, self.zdim, self.epsilon)
        mi_loss = torch.tensor(0)
        wkl = self.kl_warm_weights[epoch] if epoch < self.kl_warm_epochs else self.kl_weight
        final_loss = recon_loss + float(wkl) * kl_loss
        return final_loss, recon_loss, kl_loss, mi_loss, loss_cpx, loss_mag, sisnr
