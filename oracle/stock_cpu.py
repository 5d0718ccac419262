"""CPU baseline for bench.py: the reference's DCCRN-CL op sequence on STOCK torch operators.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (like the rest of oracle/): imported by bench.py's ``cpu_baseline`` leg and by
tests, never by the product path.

The reference runs ``torch.stft`` -> 6 x (4 x nn.Conv2d + stack, ComplexBatchNormal, nn.PReLU) -> 4 x 2-layer nn.LSTM ->
2 x nn.Linear -> 6 x (torch.cat, 4 x nn.ConvTranspose2d + stack, ComplexBatchNormal, nn.PReLU) -> mask -> ``torch.istft``
(model/pvae_module.py:174-255, model/complex_progress.py).  The oracle restates the same arithmetic with explicit loops
for the LSTM and the overlap-add, which made it ~25 % slower than the reference on the same host; this file
assembles the SAME stock modules the reference uses (MKLDNN convolutions, the fused CPU LSTM, torch.stft / istft), from
this repository's own code, so that the CPU number printed beside the GPU one is what a user of the reference would
see.  SURVEY.md 8(d) prescribes exactly this for the GPU box, where /root/reference does not exist.  The ratio
reference / this module measured in the build container is recorded in BASELINE.md.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import idccrn_oracle as O


class _CConv(nn.Module):
    def __init__(self, cin, cout, transposed, causal):
        super().__init__()
        mk = (lambda: nn.ConvTranspose2d(cin, cout, (5, 2), (2, 1), (2, 0))) if transposed else \
             (lambda: nn.Conv2d(cin, cout, (5, 2), (2, 1), (2, 1) if causal else (2, 0)))
        self.re, self.im = mk(), mk()
        self.causal = causal

    def forward(self, x):
        xr, xi = x[..., 0], x[..., 1]
        re = self.re(xr) - self.im(xi)
        im = self.re(xi) + self.im(xr)
        if self.causal:
            re, im = re[:, :, :, :-1], im[:, :, :, :-1]
        return torch.stack((re, im), dim=-1)


class _CBN(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.g_rr, self.g_ri, self.g_ii = (nn.Parameter(torch.ones(C)) for _ in range(3))
        self.b_r, self.b_i = nn.Parameter(torch.zeros(C)), nn.Parameter(torch.zeros(C))
        for n in ("mu_r", "mu_i", "Vrr", "Vri", "Vii"):
            self.register_buffer(n, torch.zeros(C))

    def forward(self, x, train):
        if train:
            st = O.cbn_batch_stats(x)
        else:
            st = (self.mu_r, self.mu_i, self.Vrr, self.Vri, self.Vii)
        return O.cbn_whiten_affine(x, *st, self.g_rr, self.g_ri, self.g_ii, self.b_r, self.b_i)


class _Block(nn.Module):
    def __init__(self, cin, cout, transposed, causal):
        super().__init__()
        self.conv = _CConv(cin, cout, transposed, causal)
        self.bn = _CBN(cout)
        self.prelu = nn.PReLU()

    def forward(self, x, train):
        return self.prelu(self.bn(self.conv(x), train))


class StockDCCRN(nn.Module):
    """DCCRN_ (causal, all skips, mask) from stock torch modules; weights copied from a reference-keyed state_dict."""

    def __init__(self, np_, n_fft, hop, win, causal=True):
        super().__init__()
        en, de = np_["encoder_channels"], np_["decoder_channels"]
        self.enc = nn.ModuleList([_Block(en[i], en[i + 1], False, causal) for i in range(len(en) - 1)])
        self.dec = nn.ModuleList([_Block(de[i] + en[len(en) - 1 - i], de[i + 1], True, causal) for i in range(len(de) - 1)])
        I, H = np_["lstm_dim"]
        self.lstm_re, self.lstm_im = nn.LSTM(I, H, 2), nn.LSTM(I, H, 2)
        self.lin_r, self.lin_i = nn.Linear(np_["dense"][0], np_["dense"][1]), nn.Linear(np_["dense"][0], np_["dense"][1])
        self.n_fft, self.hop, self.win = n_fft, hop, win
        self.register_buffer("window", torch.hann_window(win))

    @torch.no_grad()
    def load_reference_state(self, sd, prefix="std_DCCRN."):
        def blk(b, pre, conv):
            n = "tconv" if conv == "transconv" else "conv"
            for part in ("re", "im"):
                getattr(b.conv, part).weight.copy_(sd[f"{pre}{conv}.{n}_{part}.weight"])
                getattr(b.conv, part).bias.copy_(sd[f"{pre}{conv}.{n}_{part}.bias"])
            for a, k in (("g_rr", "gamma_rr"), ("g_ri", "gamma_ri"), ("g_ii", "gamma_ii"), ("b_r", "beta_r"), ("b_i", "beta_i"),
                         ("mu_r", "running_mean_real"), ("mu_i", "running_mean_imag"), ("Vrr", "Vrr"), ("Vri", "Vri"),
                         ("Vii", "Vii")):
                getattr(b.bn, a).copy_(sd[f"{pre}bn.{k}"].reshape(-1))
            b.prelu.weight.copy_(sd[f"{pre}prelu.weight"])
        for i, b in enumerate(self.enc):
            blk(b, f"{prefix}encoders.{i}.", "conv")
        for i, b in enumerate(self.dec):
            blk(b, f"{prefix}decoders.{i}.", "transconv")
        for m, name in ((self.lstm_re, "lstm_re"), (self.lstm_im, "lstm_im")):
            for k, v in m.state_dict().items():
                getattr(m, k).copy_(sd[f"{prefix}lstms.0.{name}.{k}"])
        self.lin_r.weight.copy_(sd[f"{prefix}dense.linear_read.weight"]); self.lin_r.bias.copy_(sd[f"{prefix}dense.linear_read.bias"])
        self.lin_i.weight.copy_(sd[f"{prefix}dense.linear_imag.weight"]); self.lin_i.bias.copy_(sd[f"{prefix}dense.linear_imag.bias"])
        return self

    def stft(self, x):
        return torch.view_as_real(torch.stft(x, self.n_fft, self.hop, self.win, self.window, return_complex=True))

    def forward(self, signal, train=False):
        X = self.stft(signal)                                   # [B, F, T, 2]
        x = X.unsqueeze(1)
        skips = []
        for b in self.enc:
            x = b(x, train)
            skips.append(x)
        B, C, F, T, _ = x.shape
        seq = x.reshape(B, C * F, T, 2).permute(2, 0, 1, 3)      # [T, B, C*F, 2]
        xr, xi = seq[..., 0].contiguous(), seq[..., 1].contiguous()
        rr, ri = self.lstm_re(xr)[0], self.lstm_im(xr)[0]
        ii, ir = self.lstm_im(xi)[0], self.lstm_re(xi)[0]
        lat = torch.stack((rr - ii, ir + ri), dim=-1)            # [T, B, H, 2]
        h = lat.permute(1, 0, 2, 3).reshape(B * T, -1, 2)
        d = torch.stack((self.lin_r(h[..., 0]), self.lin_i(h[..., 1])), dim=-1)
        p = d.reshape(B, T, C, F, 2).permute(0, 2, 3, 1, 4)
        for i, b in enumerate(self.dec):
            p = b(torch.cat([p, skips[len(skips) - 1 - i]], dim=1), train)
        pred = O.apply_mask(p.squeeze(1), X)
        est = torch.istft(torch.view_as_complex(pred.contiguous()), self.n_fft, self.hop, self.win, self.window)
        return est, pred
