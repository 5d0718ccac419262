"""Older non-causal DCCRN wrapper (reference: model/module.py:192-221): forward(signal, train=True) returns
the enhanced waveform only; mask output, all six skip connections, no resynthesis."""
import torch.nn as nn

from .pvae_module import DCCRN_ as _DCCRN


class DCCRN_(nn.Module):
    def __init__(self, n_fft, hop_len, net_params, device, win_length):
        super().__init__()
        core = _DCCRN(n_fft, hop_len, net_params, False, device, win_length, [0, 1, 2, 3, 4, 5], "mask", False, None, None)
        self.stft, self.DCCRN, self.istft = core.stft, core.std_DCCRN, core.istft
        self._core = [core]          # not a registered child: keys stay stft./DCCRN./istft. as in the reference

    def forward(self, signal, train=True):
        clean, _ = self._core[0](signal, train=train)
        return clean
