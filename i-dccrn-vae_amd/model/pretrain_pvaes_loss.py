"""CVAE / NVAE pre-training ELBO (reference: model/pretrain_pvaes_loss.py: KL_annealing :3-42,
complex_standard_vae_loss :48-347).  The shipped recipe runs recon_loss_type='multiple', prior_mode 'ri_inde', mi_weight 0;
'prob', 'ri_corr' and the mutual-information term (mi_weight != 0) are the class's other branches.  Reductions run in the
idv_recon_loss / idv_sisnr / idv_ckl / idv_mi_* kernels."""
import torch

from .. import autograd as AG
from .. import ops
from ..ops import Planar
from ._loss_common import kl_mean, latent_ref, recon_terms
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

    def prob_recon_loss(self, miu, input):
        """reference :161-182: the complex-spectrum term alone, mean_{b,t} sum_f |miu_x - x|^2 (predict_type 'real_imag')."""
        if self.predict_type != 'real_imag':
            raise ValueError("prob_recon_loss: predict_type 'real_imag' only (the reference leaves 'mag_wrapphase' as a TODO)")
        zero = torch.tensor(0)
        _, loss_cpx, _, _ = recon_terms(miu, input, None, None, (1.0, 0.0, 0.0), with_sisnr=False)
        return loss_cpx, zero, zero, zero

    def mutual_information(self, mu, logsigma, delta, z):
        """reference :129-159 (on cal_gaussian_prob :64-127): mean_{i,s,t} of log q(z_ist | x_i) - log mean_j q(z_ist | x_j);
        mu / logsigma / delta [B, T, H, 2], z [B * num_samples, T, H, 2] (the encoder's samples, batch-major)."""
        lat, off = latent_ref(mu, logsigma, delta)
        H = mu.shape[2]
        zp = getattr(z, "_idv", None)
        if zp is None or (zp.C, zp.Tp) != (H, lat.Tp):
            if not z.is_cuda:
                raise RuntimeError("i-dccrn-vae_amd runs on the MI355X only: pass CUDA (ROCm) tensors")
            zp = Planar.from_tensor5(z.float().permute(0, 2, 1, 3).unsqueeze(2), lat.Tp)
        if AG.grad_mode(lat.buf, zp.buf):
            return AG.MiFn.apply(AG._geom(lat), tuple(off), AG._geom(zp), H, self.num_samples, self.epsilon, lat.buf, zp.buf)
        return ops.mi_estimate(lat, off, zp, H, self.num_samples, self.epsilon)[0]

    def _prior(self, miu, log_sigma, delta):
        """:322-331: 'ri_inde' = the standard prior (None: folded into the kernel); 'ri_corr' = unit pseudo-covariance j."""
        if self.prior_mode == 'ri_inde':
            return None
        if self.prior_mode == 'ri_corr':
            delta_prior = torch.zeros_like(delta.detach())
            delta_prior[..., 1] = 1
            return torch.zeros_like(miu.detach()), torch.zeros_like(log_sigma.detach()), delta_prior
        raise ValueError(f"prior_mode {self.prior_mode!r}: the reference defines 'ri_inde' and 'ri_corr' (:322-331)")

    def cal_loss(self, source, est_source, stft_source, miu_x, miu, log_sigma, delta, z, epoch):
        """reference :313-347 -> (final, recon, kl, mi, loss_cpx, loss_mag, sisnr)"""
        if self.recon_loss_type == 'multiple':
            recon_loss, loss_cpx, loss_mag, sisnr = self.multiple_recon_loss(miu_x, stft_source, source, est_source)
        elif self.recon_loss_type == 'prob':
            recon_loss, loss_cpx, loss_mag, sisnr = self.prob_recon_loss(miu_x, stft_source)
        else:
            raise ValueError(f"recon_loss_type {self.recon_loss_type!r}: the reference defines 'multiple' and 'prob' (:316-320)")
        self.zdim = miu.shape[2]
        kl_loss = kl_mean((miu, log_sigma, delta), self._prior(miu, log_sigma, delta), self.zdim, self.epsilon)
        mi_loss = self.mutual_information(miu, log_sigma, delta, z) if self.mi_weight != 0 else torch.tensor(0)
        wkl = self.kl_warm_weights[epoch] if epoch < self.kl_warm_epochs else self.kl_weight
        final_loss = recon_loss + float(wkl) * kl_loss
        if self.mi_weight != 0:
            final_loss = final_loss - self.mi_weight * mi_loss
        return final_loss, recon_loss, kl_loss, mi_loss, loss_cpx, loss_mag, sisnr
