"""Model assembly with the reference's class names, constructor signatures, forward() signatures,
return tuples and state_dict keys (reference: model/pvae_module.py), running on the HIP kernels.

Classes: STFT, ISTFT, Encoder, Decoder, standard_DCCRN, DCCRN_ (supervised DCCRN-CL),
pvae_dccrn_encoder_skip_prepare / pvae_dccrn_decoder_skip_prepare (CVAE / NVAE pre-training),
nsvae_pvae_dccrn_encoder_twophase / nsvae_pvae_dccrn_decoder_twophase (NSVAE and decoder fine-tune).
The reference's other encoder/decoder variants are ablation copies of these.

A batch stays in the planar-J device layout from STFT to ISTFT; tensors handed back to the caller
(skiper, stft_x, z, miu, ...) are strided views with the reference's shapes, and carry their planar
buffer so that passing them into the decoder costs no copy.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .. import autograd as AG
from .. import ops
from ..ops import Planar
from .complex_progress import (ComplexBatchNormal, ComplexConv2d, ComplexConvTranspose2d, ComplexDense, ComplexLSTM,
                               causal_complex_conv2d, causal_ComplexConvTranspose2d, planar_of, tag5)

_PLANS = {}


def dft_plan(n_fft, win, hop, T, device) -> ops.DftPlan:
    key = (n_fft, win, hop, T, str(device))
    if key not in _PLANS:
        _PLANS[key] = ops.DftPlan(n_fft, win, hop, T, device)
        torch.cuda.current_stream(device).synchronize()       # built once; users may sit on other streams
    return _PLANS[key]


def _need_cuda(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("i-dccrn-vae_amd runs on the MI355X only: pass CUDA (ROCm) tensors; there is no CPU path")


class STFT(nn.Module):
    """reference: model/pvae_module.py:12-27 (torch.stft, hann(win), center/reflect) -> [B, F, T, 2]"""

    def __init__(self, n_fft, hop_length, win_length, device):
        super().__init__()
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length

    def planar(self, signal: torch.Tensor) -> Planar:
        _need_cuda(signal)
        T = 1 + signal.shape[1] // self.hop_length
        plan = dft_plan(self.n_fft, self.win_length, self.hop_length, T, signal.device)
        if AG.grad_mode(signal):
            sig = signal.float().contiguous()
            buf = AG.StftFn.apply(plan, sig)
            return Planar(buf, 1, plan.F, sig.shape[0], T, T + 1, Planar.jp_for(sig.shape[0], T + 1))
        return ops.stft(signal.float(), plan)

    def forward(self, signal):
        pl = self.planar(signal)
        x = pl.tensor4()
        x._idv = pl
        return x


class ISTFT(nn.Module):
    """reference: model/pvae_module.py:30-42 (torch.istft) : complex [B, F, T] -> [B, hop*(T-1)]"""

    def __init__(self, n_fft, hop_length, win_length, device):
        super().__init__()
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length

    def planar(self, spec: Planar) -> torch.Tensor:
        plan = dft_plan(self.n_fft, self.win_length, self.hop_length, spec.T, spec.buf.device)
        if AG.grad_mode(spec.buf):
            return AG.IstftFn.apply(plan, AG._geom(spec), spec.buf)
        return ops.istft(spec, plan)

    def forward(self, x):
        pl = getattr(x, "_idv", None)
        if pl is None:
            _need_cuda(x)
            pl = Planar.from_tensor5(torch.view_as_real(x).unsqueeze(1).float())
        return self.planar(pl)


class _Block(nn.Module):
    """conv -> ComplexBatchNormal -> PReLU.  Eval mode: the running-statistics affine is folded into the
    packed weights and PReLU runs in the conv epilogue (one kernel).  Train mode: the conv epilogue
    emits the batch moments, then one finalise and one in-place normalise+PReLU kernel."""

    def _run(self, conv, x: Planar, train: bool, **kw) -> Planar:
        slope = self.prelu.weight.detach()
        skip = kw.get("skip")
        if train and AG.grad_mode(x.buf, skip.buf if skip is not None else None, *self.parameters()):
            # training step: autograd.Function with the HIP backward kernels (conv + train-mode CBN + PReLU)
            if kw.get("want", "planar") != "planar":
                raise RuntimeError("split images are an eval-mode format")
            div = kw.get("skip_div", 1)
            if skip is not None and div > 1:
                if AG.skip_once_ok(conv, x, skip, div):      # fp32: skip half of the conv once per utterance
                    return AG.conv_block(conv, self.bn, self.prelu.weight, x, skip, kw.get("zero_skip", False), skip_div=div)
                skip = AG.repeat_batch(skip, div)
            return AG.conv_block(conv, self.bn, self.prelu.weight, x, skip, kw.get("zero_skip", False))
        if train:
            if kw.get("want", "planar") != "planar":
                raise RuntimeError("split images are an eval-mode format (train-mode CBN normalises planar fp32 in place)")
            stats = torch.zeros(conv.out_channel, 5, dtype=torch.float64, device=x.buf.device)
            y = conv.forward_planar(x, stats=stats, **kw)
            return self.bn.finish_train(y, stats, slope)
        return conv.forward_planar(x, fold=self.bn.eval_fold(), slope=slope, **kw)


class Encoder(_Block):
    """reference: model/pvae_module.py:45-68"""

    def __init__(self, in_channel, out_channel, kernel_size, stride, chw, padding=None, causal=False):
        super().__init__()
        if padding is None:
            padding = [int((k - 1) / 2) for k in kernel_size]
        cls = causal_complex_conv2d if causal else ComplexConv2d
        self.conv = cls(in_channel=in_channel, out_channel=out_channel, kernel_size=kernel_size, stride=stride,
                        padding=padding)
        self.bn = ComplexBatchNormal(chw[0], chw[1], chw[2])
        self.prelu = nn.PReLU()

    def forward_planar(self, x, train: bool, want: str = "planar"):
        return self._run(self.conv, x, train, want=want)

    def forward(self, x, train):
        return tag5(self.forward_planar(planar_of(x), train))


class Decoder(_Block):
    """reference: model/pvae_module.py:72-93"""

    def __init__(self, in_channel, out_channel, kernel_size, stride, chw, padding=None, causal=False, if_bn=True):
        super().__init__()
        cls = causal_ComplexConvTranspose2d if causal else ComplexConvTranspose2d
        self.transconv = cls(in_channel=in_channel, out_channel=out_channel, kernel_size=kernel_size, stride=stride,
                             padding=padding)
        self.bn = ComplexBatchNormal(chw[0], chw[1], chw[2])
        self.prelu = nn.PReLU()
        self.if_bn = if_bn

    def forward_planar(self, x, train: bool = True, skip=None, skip_div: int = 1,
                       zero_skip: bool = False, want: str = "planar"):
        kw = dict(skip=skip, skip_div=skip_div, zero_skip=zero_skip, want=want)
        if not self.if_bn:
            if AG.grad_mode(x.buf, skip.buf if skip is not None else None, *self.transconv.parameters()):
                if skip is not None and skip_div > 1:
                    skip = AG.repeat_batch(skip, skip_div)
                return AG.conv_block(self.transconv, None, None, x, skip, zero_skip)
            return self.transconv.forward_planar(x, **kw)
        return self._run(self.transconv, x, train, **kw)

    def forward(self, x, train=True):
        return tag5(self.forward_planar(planar_of(x), train))


def _build_encoders(net_params, causal) -> nn.ModuleList:
    ch = net_params["encoder_channels"]
    return nn.ModuleList([
        Encoder(in_channel=ch[i], out_channel=ch[i + 1], kernel_size=net_params["encoder_kernel_sizes"][i],
                stride=net_params["encoder_strides"][i], padding=net_params["encoder_paddings"][i],
                chw=net_params["encoder_chw"][i], causal=causal)
        for i in range(len(ch) - 1)])


def _build_decoders(net_params, causal, skip_to_use, use_sc=True) -> nn.ModuleList:
    en, de = net_params["encoder_channels"], net_params["decoder_channels"]
    blocks = []
    for i in range(len(de) - 1):
        cin = de[i] + (en[len(en) - 1 - i] if (use_sc and i in skip_to_use) else 0)
        blocks.append(Decoder(in_channel=cin, out_channel=de[i + 1], kernel_size=net_params["decoder_kernel_sizes"][i],
                              stride=net_params["decoder_strides"][i], padding=net_params["decoder_paddings"][i],
                              chw=net_params["decoder_chw"][i], causal=causal))
    return nn.ModuleList(blocks)


def _run_encoders(encoders, x: Planar, train: bool) -> List[Planar]:
    outs = []
    images = (not train) and ops.PRECISION == "bf16x3" and ops.IMAGE_PATH
    for i, enc in enumerate(encoders):
        nxt = encoders[i + 1].conv if i + 1 < len(encoders) else None
        if images and nxt is not None and nxt.takes_images(enc.conv.out_channel, 0):
            # eval bf16x3: the planar output is what the caller gets (skiper), the split image feeds the next block
            p, x = enc.forward_planar(x, False, want="both")
            outs.append(p)
        else:
            x = enc.forward_planar(x, train)
            outs.append(x)
    return outs


def _lstm_out_view(lat: Planar, c0: int, c1: int) -> torch.Tensor:
    v = lat.channel_slice(c0, c1)
    v._idv = lat
    v._idv_off = c0
    return v


class standard_DCCRN(nn.Module):
    """reference: model/pvae_module.py:96-198"""

    def __init__(self, net_params, causal, device, skip_to_use):
        super().__init__()
        self.device = device
        self.causal = causal
        self.skip_to_use = skip_to_use
        self.dense = ComplexDense(net_params["dense"][0], net_params["dense"][1])
        self.encoders = _build_encoders(net_params, causal)
        dims = net_params["lstm_dim"]
        self.lstms = nn.ModuleList([
            ComplexLSTM(input_size=dims[i], hidden_size=dims[i + 1], num_layers=net_params["lstm_layer_num"], device=device)
            for i in range(len(dims) - 1)])
        self.decoders = _build_decoders(net_params, causal, skip_to_use)
        self.linear = ComplexConv2d(in_channel=1, out_channel=1, kernel_size=1, stride=1)   # unused, kept for state_dict parity
        self.detect_anormal = True

    def forward_planar(self, x: Planar, train: bool = True, on_encoded=None) -> Planar:
        if not train and ops.PRECISION == "bf16x3" and ops.IMAGE_PATH:
            return self._forward_images(x, on_encoded)
        skips = _run_encoders(self.encoders, x, train)
        if on_encoded is not None:
            on_encoded()
        top = skips[-1]
        lat = top
        for lstm in self.lstms:
            lat = lstm.forward_planar(lat)
        if not train:
            self.latent = _lstm_out_view(lat, 0, lat.C)            # [B, T, H, 2]
        p = self.dense.forward_planar(lat, top.C, top.F)
        for i, dec in enumerate(self.decoders):
            p = dec.forward_planar(p, train, skip=skips[len(skips) - 1 - i] if i in self.skip_to_use else None)
        return p

    def _forward_images(self, x: Planar, on_encoded=None) -> Planar:
        """Eval bf16x3: activations between conv blocks live as split-bf16 images (ops.Image) that the next kernel
        copies into LDS verbatim; planar fp32 only where a non-conv consumer needs it (LSTM input, the Cout = 1
        last block and its skip, the mask).  Same arithmetic as the planar bf16x3 path, bit for bit."""
        n = len(self.encoders)
        skips = []                                                  # per encoder: what the decoder side will read
        top = None
        for i, enc in enumerate(self.encoders):
            last = i == n - 1
            nxt_needs_planar = last                                 # LSTM projection reads planar fp32
            dec_i = n - 1 - i                                       # decoder that takes this output as its skip
            dtc = self.decoders[dec_i].transconv
            skip_planar = dec_i in self.skip_to_use and not dtc.takes_images(dtc.in_channel - enc.conv.out_channel,
                                                                             enc.conv.out_channel)
            if nxt_needs_planar or skip_planar:
                want = "both"
            else:
                want = "image"
            y = enc.forward_planar(x, False, want=want)
            if want == "both":
                top, x = y
                skips.append(y[0] if skip_planar else y[1])
            else:
                top = x = y
                skips.append(y)
        if on_encoded is not None:
            on_encoded()
        if not isinstance(top, Planar):
            top = ops.to_planar(top)
        lat = top
        for lstm in self.lstms:
            lat = lstm.forward_planar(lat)
        self.latent = _lstm_out_view(lat, 0, lat.C)                # [B, T, H, 2]
        p = self.dense.forward_planar(lat, top.C, top.F)
        nd = len(self.decoders)
        for i, dec in enumerate(self.decoders):
            nxt = self.decoders[i + 1].transconv if i + 1 < nd else None
            c_out = dec.transconv.out_channel
            want = "image" if (nxt is not None and nxt.takes_images(c_out, nxt.in_channel - c_out)) else "planar"
            p = dec.forward_planar(p, False, skip=skips[n - 1 - i] if i in self.skip_to_use else None, want=want)
        return p

    def forward(self, x, train=True):
        return tag5(self.forward_planar(planar_of(x), train))


def _apply_datanorm(stft: Planar, mean, std, train: bool = False) -> Planar:
    """Optional input normalisation of DCCRN_.forward (pvae_module.py:217-221); off in the shipped recipes."""
    if train and torch.is_grad_enabled():
        buf = AG.DatanormFn.apply(AG._geom(stft), mean.reshape(-1).float().contiguous(), std.reshape(-1).float().contiguous(), stft.buf)
        return ops.rewrap(buf, stft)
    out = Planar.empty(1, stft.F, stft.B, stft.T, stft.Tp, stft.buf.device)
    ops.call("idv_datanorm", stft.ptr(), ops.p(mean.reshape(-1).float().contiguous()), ops.p(std.reshape(-1).float().contiguous()),
             ops.i(stft.F), ops.i(stft.B), ops.i(stft.T), ops.i(stft.Tp), ops.i(stft.Jp), out.ptr(), ops.stream_ptr())
    return out


def _invert_datanorm(pred: Planar, mean, std):
    """predict = data_std * predict + data_mean (pvae_module.py:235-238) -> (planar, complex [B, F, T])."""
    if AG.grad_mode(pred.buf):
        obuf, pc = AG.DatadenormFn.apply(AG._geom(pred), mean.reshape(-1).float().contiguous(), std.reshape(-1).float().contiguous(),
                                         pred.buf)
        return ops.rewrap(obuf, pred), torch.view_as_complex(pc)
    out = Planar.empty(1, pred.F, pred.B, pred.T, pred.Tp, pred.buf.device)
    pc = torch.empty(pred.B, pred.F, pred.T, 2, dtype=torch.float32, device=pred.buf.device)
    ops.call("idv_datadenorm", pred.ptr(), ops.p(mean.reshape(-1).float().contiguous()), ops.p(std.reshape(-1).float().contiguous()),
             ops.i(pred.F), ops.i(pred.B), ops.i(pred.T), ops.i(pred.Tp), ops.i(pred.Jp), out.ptr(), ops.p(pc), ops.stream_ptr())
    return out, torch.view_as_complex(pc)


def _predict_outputs(module, out: Planar, stft_in: Planar, recon_type: str, x_div: int = 1):
    """Mask / real_imag branch shared by DCCRN_ and the decoders -> (pred planar, predict complex [B,F,T])."""
    if recon_type == "mask":
        if AG.grad_mode(out.buf, stft_in.buf if x_div == 1 else None):
            xb = stft_in.buf if x_div == 1 else stft_in.buf.detach()
            pbuf, pcr = AG.MaskFn.apply(AG._geom(out), AG._geom(stft_in), x_div, out.buf, xb)
            return ops.rewrap(pbuf, out), torch.view_as_complex(pcr)
        return ops.mask_apply(out, stft_in, x_div)
    if recon_type == "real_imag":
        if AG.grad_mode(out.buf):
            return out, torch.view_as_complex(AG.PlanarToComplexFn.apply(AG._geom(out), out.buf))
        return out, ops.planar_to_complex(out)
    raise ValueError(f"recon_type {recon_type!r}")


def _eval_builds_no_graph(what, *inputs):
    """train=False runs the folded-BN eval kernels, which have no backward.  The reference would still build a graph
    through an eval-mode module; silently cutting a gradient that something upstream needs would be wrong, so an input
    that requires grad is refused instead (parameters of the module itself get no gradient from an eval-mode forward)."""
    def walk(t):
        if isinstance(t, torch.Tensor):
            return t.requires_grad
        if isinstance(t, (list, tuple)):
            return any(walk(u) for u in t)
        return False
    if any(walk(t) for t in inputs):
        raise NotImplementedError(
            f"{what}: forward(train=False) under enabled grad with an input that requires grad -- the eval-mode (folded batch norm) "
            "kernels have no backward, the gradient would be cut silently; call with train=True, or detach() the input")


class DCCRN_(nn.Module):
    """Supervised DCCRN (DCCRN-CL when causal=True).  reference: model/pvae_module.py:200-255.
    forward(signal [B, L], train=True) -> (clean [B, hop*(T-1)], predict complex64 [B, F, T])."""

    def __init__(self, n_fft, hop_len, net_params, causal, device, win_length, skip_to_use, recon_type, resynthesis,
                 data_mean, data_std):
        super().__init__()
        self.stft = STFT(n_fft, hop_len, win_length=win_length, device=device)
        self.std_DCCRN = standard_DCCRN(net_params, causal, device=device, skip_to_use=skip_to_use)
        self.istft = ISTFT(n_fft, hop_len, win_length=win_length, device=device)
        self.recon_type = recon_type
        self.resynthesis = resynthesis
        self.register_buffer("data_mean", data_mean)
        self.register_buffer("data_std", data_std)
        self.datanorm = self.data_mean is not None and self.data_std is not None

    def forward(self, signal, train=True):
        # Eval is per-utterance independent (folded BN): run sub-batches on separate HIP streams so one sub-batch's
        # 16-CU LSTM recurrence and kernel tails overlap the other's conv GEMMs.  Train mode needs whole-batch
        # CBN statistics and stays on one stream.
        if not train and torch.is_grad_enabled():
            # eval mode builds no graph: the folded-BN kernels have no backward (module docstring)
            _eval_builds_no_graph(type(self).__name__, signal)
            with torch.no_grad():
                return self.forward(signal, False)
        n = 1 if train else ops.stream_split(signal.shape[0])
        if n == 1:
            return self._forward_one(signal, train)
        main = torch.cuda.current_stream(signal.device)
        outs = []
        streams = ops.side_streams(n, signal.device)
        ready = torch.cuda.Event()
        ready.record(main)
        encoded = None                                            # staggered: part k+1 starts when part k enters its LSTM
        for part, st in zip(signal.tensor_split(n), streams):
            st.wait_event(ready)
            part.record_stream(st)                                # the input is read on st: its block is not re-used before st is done
            if encoded is not None and ops.STREAM_STAGGER and part.shape[0] < ops.STREAM_STAGGER_BELOW:
                st.wait_event(encoded)
            encoded = torch.cuda.Event()
            with torch.cuda.stream(st):
                outs.append(self._forward_one(part, False, encoded.record) + (self.std_DCCRN.latent,))
        for st in streams:                                        # join only after every part is enqueued
            main.wait_stream(st)
        # Explicit cross-stream ownership: every part's outputs were allocated from their side stream's pool
        # and are read by the concatenations on `main`; record_stream makes the allocator keep those blocks until main has
        # passed the point where they are released, whatever the side stream does next.
        ops.hand_over(outs, main)
        clean, predict, latent = (torch.cat([o[k] for o in outs]) for k in range(3))
        self.std_DCCRN.latent = latent
        return clean, predict

    def _forward_one(self, signal, train, on_encoded=None):
        X = self.stft.planar(signal)
        net_in = _apply_datanorm(X, self.data_mean, self.data_std, train) if self.datanorm else X
        out = self.std_DCCRN.forward_planar(net_in, train=train, on_encoded=on_encoded)
        pred, predict = _predict_outputs(self, out, net_in, self.recon_type)
        if self.datanorm:
            pred, predict = _invert_datanorm(pred, self.data_mean, self.data_std)
        clean = self.istft.planar(pred)
        if self.resynthesis:
            predict = ops.planar_to_complex(self.stft.planar(clean))
        else:
            predict._idv = pred
        return clean, predict


class _VAEEncoderBase(nn.Module):
    """STFT -> 6 encoder blocks -> ComplexLSTM(1280 -> 3*zdim*latent_num) -> (miu | log_sigma | delta) per latent
    -> reparameterised samples."""

    def _setup(self, net_params, causal, device, zdim, n_fft, hop_len, win_length, num_samples, latent_num):
        self.device = device
        self.causal = causal
        self.stft = STFT(n_fft, hop_len, win_length=win_length, device=device)
        self.dense = ComplexDense(zdim, net_params["dense"][1])    # constructed and unused in the reference too
        self.decoders = []
        self.zdim = zdim
        self.num_samples = num_samples
        self.encoders = _build_encoders(net_params, causal)
        dims = net_params["lstm_dim"]
        hidden = int(3 * zdim * latent_num)
        self.lstms = nn.ModuleList([
            ComplexLSTM(input_size=dims[i], hidden_size=hidden, num_layers=net_params["lstm_layer_num"], device=device)
            for i in range(len(dims) - 1)])
        self.epsilon = 1e-6

    def _sample(self, lat: Planar, k: int, eps) -> torch.Tensor:
        """reparameterization (pvae_module.py:1832-1886) for latent k; eps = (eps_r, eps_i) or None."""
        z = self.zdim
        B, T, ns = lat.B, lat.T, self.num_samples
        if eps is None:
            eps = (torch.randn(B, ns, T, z, device=lat.buf.device), torch.randn(B, ns, T, z, device=lat.buf.device))
        off = (3 * z * k, 3 * z * k + z, 3 * z * k + 2 * z)
        if AG.grad_mode(lat.buf):
            zbuf = AG.ReparamFn.apply(AG._geom(lat), off, z, ns, lat.buf, eps[0], eps[1])
            zp = Planar(zbuf, z, 1, B * ns, T, lat.Tp, Planar.jp_for(B * ns, lat.Tp))
        else:
            zp = ops.reparam(lat, off, z, eps[0], eps[1], ns)
        v = zp.channel_slice(0, z)                                  # [B*ns, T, zdim, 2]
        v._idv = zp
        return v

    def reparameterization(self, miu, log_sigma, delta, num_samples, eps=None):
        """Stand-alone form with the reference's signature: miu/log_sigma/delta are [B, T, H, 2]."""
        lat = torch.cat([miu, log_sigma, delta], dim=2).permute(0, 2, 1, 3).unsqueeze(2)
        pl = Planar.from_tensor5(lat.float())
        keep_ns, keep_z = self.num_samples, self.zdim
        self.num_samples, self.zdim = num_samples, miu.shape[2]
        try:
            return self._sample(pl, 0, eps)
        finally:
            self.num_samples, self.zdim = keep_ns, keep_z

    def _encode(self, x, train):
        X = self.stft.planar(x)
        skips = _run_encoders(self.encoders, X, train)
        top = skips[-1]
        lat = top
        for lstm in self.lstms:
            lat = lstm.forward_planar(lat)
        stft_x = X.tensor4()
        stft_x._idv = X
        return lat, [tag5(s) for s in skips], top.C, top.F, stft_x


class pvae_dccrn_encoder_skip_prepare(_VAEEncoderBase):
    """CVAE / NVAE encoder.  reference: model/pvae_module.py:1791-1914.
    forward(x, train=True, eps=None) -> (z, miu, log_sigma, delta, skiper, C, F, stft_x);
    ``eps=(eps_r, eps_i)`` ([B, ns, T, zdim]) injects the two Gaussian draws (parity tests), default samples on device."""

    def __init__(self, net_params, causal, device, zdim, n_fft, hop_len, win_length, num_samples):
        super().__init__()
        self._setup(net_params, causal, device, zdim, n_fft, hop_len, win_length, num_samples, 1)

    def forward(self, x, train=True, eps=None):
        if not train and torch.is_grad_enabled():
            _eval_builds_no_graph(type(self).__name__, x)
            with torch.no_grad():                                    # eval mode builds no graph
                return self.forward(x, False, eps)
        lat, skiper, C, F, stft_x = self._encode(x, train)
        z = self.zdim
        return (self._sample(lat, 0, eps), _lstm_out_view(lat, 0, z), _lstm_out_view(lat, z, 2 * z),
                _lstm_out_view(lat, 2 * z, 3 * z), skiper, C, F, stft_x)


class nsvae_pvae_dccrn_encoder_twophase(_VAEEncoderBase):
    """Noisy-speech (NSVAE) encoder.  reference: model/pvae_module.py:2131-2268.  forward -> 12-tuple
    (z_speech, miu_speech, log_sigma_speech, delta_speech, z_noise, miu_noise, log_sigma_noise, delta_noise,
    skiper, C, F, stft_x); eps = (eps_r_speech, eps_i_speech[, eps_r_noise, eps_i_noise])."""

    def __init__(self, net_params, causal, device, zdim, n_fft, hop_len, win_length, num_samples, latent_num):
        super().__init__()
        if latent_num not in (1, 2):
            raise ValueError("latent_num must be 1 or 2")
        self.latent_num = latent_num
        self._setup(net_params, causal, device, zdim, n_fft, hop_len, win_length, num_samples, latent_num)

    def forward(self, x, train=True, eps=None):
        if not train and torch.is_grad_enabled():
            _eval_builds_no_graph(type(self).__name__, x)
            with torch.no_grad():                                    # eval mode builds no graph
                return self.forward(x, False, eps)
        lat, skiper, C, F, stft_x = self._encode(x, train)
        z = self.zdim
        out = []
        for k in range(2):
            if k < self.latent_num:
                e = None if eps is None else (eps[2 * k], eps[2 * k + 1])
                o = 3 * z * k
                out += [self._sample(lat, k, e), _lstm_out_view(lat, o, o + z), _lstm_out_view(lat, o + z, o + 2 * z),
                        _lstm_out_view(lat, o + 2 * z, o + 3 * z)]
            else:
                out += [None, None, None, None]
        return (*out, skiper, C, F, stft_x)


class _VAEDecoderBase(nn.Module):
    def _setup(self, net_params, causal, device, num_samples, zdim, n_fft, hop_len, win_length, recon_type, skip_to_use,
               use_sc=True):
        self.device = device
        self.causal = causal
        self.num_samples = num_samples
        self.zdim = zdim
        self.recon_type = recon_type
        self.skip_to_use = skip_to_use
        self.use_sc = use_sc
        self.dense = ComplexDense(zdim, net_params["dense"][1])
        self.decoders = _build_decoders(net_params, causal, skip_to_use, use_sc)
        self.istft = ISTFT(n_fft, hop_len, win_length=win_length, device=device)

    def _decode(self, stft_x, z, skiper, C, F, train, pad):
        if not train and torch.is_grad_enabled():
            _eval_builds_no_graph(type(self).__name__, z, skiper if pad != 'zero' else None)
            with torch.no_grad():                                    # eval mode builds no graph
                return self._decode(stft_x, z, skiper, C, F, False, pad)
        zp = getattr(z, "_idv", None)
        if zp is None:
            _need_cuda(z)
            zp = Planar.from_tensor5(z.permute(0, 2, 1, 3).unsqueeze(2).float())     # [Bn, zdim, 1, T, 2]
        Bn = zp.B
        p = self.dense.forward_planar(zp, C, F)
        outs = []
        # eval bf16x3 without real skips: the blocks hand split-bf16 images to each other (see standard_DCCRN)
        images = (not train) and ops.PRECISION == "bf16x3" and ops.IMAGE_PATH
        if pad not in ("zero", "sig"):
            raise ValueError(f"pad {pad!r}")
        nd = len(self.decoders)
        for i, dec in enumerate(self.decoders):
            if images:
                has_skip = self.use_sc and i in self.skip_to_use
                sk = None
                if has_skip and pad == "sig":
                    # repeated skips: materialise the repeat once (a copy of <= 2 x 22 MB per utterance) so the block
                    # runs on the split-image kernels instead of the exact-fp32 scalar-staged path
                    sk = planar_of(skiper[len(skiper) - 1 - i], zp.Tp)
                    if Bn % sk.B:
                        raise RuntimeError("batch of z is not a multiple of the skip batch")
                    if Bn != sk.B:
                        # behind a block that handed over an image the skip is lifted anyway: repeat inside that conversion
                        if isinstance(p, ops.Image) and dec.transconv.takes_images(p.C, sk.C):
                            sk = ops.to_image_repeat(sk, Bn // sk.B)
                        else:
                            sk = ops.repeat_batch(sk, Bn // sk.B)
                nxt = self.decoders[i + 1].transconv if i + 1 < nd else None
                c_out = dec.transconv.out_channel
                nxt_skip = 0
                if nxt is not None and pad == "sig" and self.use_sc and (i + 1) in self.skip_to_use:
                    nxt_skip = nxt.in_channel - c_out
                want = "image" if (nxt is not None and nxt.takes_images(c_out, nxt_skip)) else "planar"
                p = dec.forward_planar(p, False, skip=sk, zero_skip=has_skip and pad == "zero", want=want)
                outs.append(p)
                continue
            if self.use_sc and i in self.skip_to_use:
                if pad == "zero":
                    p = dec.forward_planar(p, train, zero_skip=True)       # cat with zeros == half of K skipped
                elif pad == "sig":
                    sk = planar_of(skiper[len(skiper) - 1 - i], zp.Tp)
                    if Bn % sk.B:
                        raise RuntimeError("batch of z is not a multiple of the skip batch")
                    p = dec.forward_planar(p, train, skip=sk, skip_div=Bn // sk.B)
                else:
                    raise ValueError(f"pad {pad!r}")
            else:
                p = dec.forward_planar(p, train)
            outs.append(p)
        X = planar_of(stft_x.unsqueeze(1), zp.Tp) if getattr(stft_x, "_idv", None) is None else stft_x._idv
        pred, predict = _predict_outputs(self, p, X, self.recon_type, x_div=Bn // X.B)
        recon = self.istft.planar(pred)
        predict._idv = pred
        return recon, predict, outs


class pvae_dccrn_decoder_skip_prepare(_VAEDecoderBase):
    """CVAE / NVAE decoder (skip inputs replaced by zeros).  reference: model/pvae_module.py:2045-2122.
    forward(stft_x, z, skiper, C, F, train=True) -> (recon_sig [B*ns, L'], predict complex [B*ns, F, T])."""

    def __init__(self, net_params, causal, device, num_samples, zdim, n_fft, hop_len, win_length, recon_type, skip_to_use):
        super().__init__()
        self._setup(net_params, causal, device, num_samples, zdim, n_fft, hop_len, win_length, recon_type, skip_to_use)

    def forward(self, stft_x, z, skiper, C, F, train=True):
        if self.recon_type != "real_imag":
            raise ValueError("pvae_dccrn_decoder_skip_prepare implements recon_type='real_imag' (as the reference)")
        recon, predict, outs = self._decode(stft_x, z, skiper, C, F, train, "zero")
        self._decoder_outs = outs
        return recon, predict

    @property
    def decoder_outputs(self):
        """Per-block outputs [B*ns, C, F, T, 2] (reference: self.decoder_outputs, pvae_module.py:2090,:2099); blocks that
        handed a split image to the next one are decoded on access."""
        return [tag5(o if isinstance(o, Planar) else ops.to_planar(o)) for o in getattr(self, "_decoder_outs", [])]


class nsvae_pvae_dccrn_decoder_twophase(_VAEDecoderBase):
    """Fine-tuned CVAE decoder.  reference: model/pvae_module.py:2505-2619.
    forward(stft_x, z, skiper, C, F, train=True, pad='zero'|'sig') -> (recon_sig, predict)."""

    def __init__(self, net_params, causal, device, num_samples, zdim, n_fft, hop_len, win_length, recon_type, use_sc,
                 skip_to_use, resynthesis):
        super().__init__()
        self._setup(net_params, causal, device, num_samples, zdim, n_fft, hop_len, win_length, recon_type, skip_to_use,
                    use_sc)
        self.resynthesis = resynthesis
        self.stft = STFT(n_fft, hop_len, win_length=win_length, device=device)

    def forward(self, stft_x, z, skiper, C, F, train=True, pad="zero"):
        recon, predict, _ = self._decode(stft_x, z, skiper, C, F, train, pad)
        if self.resynthesis:
            predict = ops.planar_to_complex(self.stft.planar(recon))
        return recon, predict
