"""SI-SNR loss (reference: model/sisnr_loss.py:7-25), computed by the idv_sisnr HIP kernel as three dot
products per utterance instead of the reference's BxB matmul."""
import torch

from .. import autograd as AG
from .. import ops


def si_snr(source, estimate_source, eps=1e-8):
    if eps != 1e-8:
        raise NotImplementedError("the HIP kernel uses the reference's eps = 1e-8")
    source = source.squeeze(1) if source.dim() == 3 else source
    estimate_source = estimate_source.squeeze(1) if estimate_source.dim() == 3 else estimate_source
    if not estimate_source.is_cuda:
        raise RuntimeError("i-dccrn-vae_amd runs on the MI355X only: pass CUDA (ROCm) tensors")
    ops.check_dev_f32(source, "source", estimate_source.device)
    if source.dim() != 2 or estimate_source.dim() != 2 or source.shape[1] < estimate_source.shape[1]:
        raise RuntimeError(f"si_snr: source {tuple(source.shape)} must cover the estimate {tuple(estimate_source.shape)}")
    src_div = 1
    if source.shape[0] != estimate_source.shape[0]:
        src_div = estimate_source.shape[0] // source.shape[0]
        if src_div * source.shape[0] != estimate_source.shape[0]:
            raise RuntimeError("si_snr: estimate batch must be a multiple of the source batch")
    if AG.grad_mode(estimate_source):
        return AG.SisnrFn.apply(_rows(source.detach()), _rows(estimate_source), src_div)
    return ops.sisnr(_rows(source), _rows(estimate_source), src_div)


def _rows(t):
    t = t.float()
    return t if t.stride(-1) == 1 else t.contiguous()


class SiSnr(object):
    def __call__(self, source, estimate_source):
        return si_snr(source, estimate_source)
