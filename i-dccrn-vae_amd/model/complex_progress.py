"""Complex-valued layers with the reference's class names, constructor arguments, parameter /
buffer names (state_dict keys) and forward() signatures (reference: model/complex_progress.py),
executed by the HIP kernels of libidccrn_hip.so.

The nn.Conv2d / nn.ConvTranspose2d / nn.LSTM / nn.Linear children are parameter containers
only (same shapes, names and default initialisation as the reference); their torch forward is
never called.  Every layer has two faces:

* ``forward(x)`` with reference-layout tensors ``[B, C, F, T, 2]`` (drop-in use), and
* ``forward_planar(...)`` on :class:`ops.Planar` activations, which the model classes chain so
  that an utterance batch stays in the planar-J device layout from STFT to ISTFT.

Training: with ``train=True`` under ``torch.enable_grad()`` the layers run through the ``torch.autograd.Function``
classes of ``autograd.py`` (HIP backward kernels), so ``loss.backward()`` fills ``.grad`` of the parameters exactly as
in the reference's train steps; eval-mode outputs carry no graph.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import autograd as AG
from .. import ops
from ..ops import Planar


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def planar_of(x: torch.Tensor, Tp: Optional[int] = None) -> Planar:
    """The planar activation behind a reference-layout tensor: the attached one when the tensor was
    produced by this package, otherwise a converted copy (device-side torch copy, no arithmetic)."""
    pl = getattr(x, "_idv", None)
    if pl is not None and (Tp is None or pl.Tp == Tp):
        return pl
    if not x.is_cuda:
        raise RuntimeError("i-dccrn-vae_amd runs on the MI355X only: pass CUDA (ROCm) tensors; there is no CPU path")
    return Planar.from_tensor5(x.float(), Tp)


def tag5(pl: Planar) -> torch.Tensor:
    t = pl.tensor5()
    t._idv = pl
    return t


def invalidate_caches(module: nn.Module):
    """Drop every packed-weight / folded-BN cache below `module`.  The caches key on (data_ptr, _version), which covers
    optimizer steps, load_state_dict and in-place edits of the parameters themselves; writes that bypass the version
    counter (``p.data.copy_(...)``, raw-pointer kernels) need this call."""
    for m in module.modules():
        for v in vars(m).values():
            if isinstance(v, _PackCache):
                v.key = None


def _tensors_of(v):
    if isinstance(v, torch.Tensor):
        yield v
    elif isinstance(v, (tuple, list)):
        for u in v:
            yield from _tensors_of(u)


class _PackCache:
    """Re-pack weights only when a parameter changed (tensor._version / storage) or the fold did.

    Cross-stream ownership (sub-batch streams of DCCRN_.forward, ops.concurrent) is explicit:
    * the pack kernels run on the stream that found the key changed; `ready` is recorded there and EVERY other stream waits
      on it on the device before its first use of this value (one wait per stream, remembered in `waited`; the event is
      kept for the life of the value, a completed event costs nothing to wait on);
    * every stream that reads the value is told to the caching allocator (`record_stream`), so that when the value is
      replaced its memory is not handed out again before the work those streams had queued on it has finished."""

    def __init__(self):
        self.key = None
        self.val = None
        self.ready = None
        self.owner = None            # stream (handle) the value was packed on
        self.waited = set()          # streams that are ordered behind `ready`, and known to the allocator as users

    def get(self, tensors, extra, build):
        key = tuple((t.data_ptr(), t._version) for t in tensors if t is not None) + (extra,)
        cur = torch.cuda.current_stream()
        if key != self.key:
            for t in tensors:
                # the pack kernels dereference raw pointers: parameters left on the CPU (module built but never moved) must
                # fail here with a message, not as a GPU memory fault
                if t is not None and not t.is_cuda:
                    raise RuntimeError("i-dccrn-vae_amd: a parameter of this module is on the CPU; move the module to the GPU "
                                       "(module.cuda()) -- the HIP hot path has no CPU fallback")
            self.val = build()       # the old value's blocks go back to the allocator, which holds them for recorded users
            self.key = key
            self.ready = torch.cuda.Event()
            self.ready.record(cur)
            self.owner = cur.cuda_stream
            self.waited = {self.owner}
        elif cur.cuda_stream not in self.waited:
            cur.wait_event(self.ready)
            for t in _tensors_of(self.val):
                t.record_stream(cur)
            self.waited.add(cur.cuda_stream)
        return self.val


class _ComplexConvBase(nn.Module):
    _transposed = False
    _causal = False
    _names = ("conv_re", "conv_im")

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, output_padding=0, dilation=1,
                 groups=1, bias=True):
        super().__init__()
        kw = dict(kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self._transposed:
            kw["output_padding"] = output_padding
            mk = lambda: nn.ConvTranspose2d(in_channel, out_channel, **kw)
        else:
            mk = lambda: nn.Conv2d(in_channel, out_channel, **kw)
        setattr(self, self._names[0], mk())
        setattr(self, self._names[1], mk())
        self.in_channel, self.out_channel = in_channel, out_channel
        self._cfg = (_pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation), groups, bias, _pair(output_padding))
        self._cache = _PackCache()
        self._cache_bf16 = _PackCache()
        self._cache_c1 = _PackCache()
        self._cache_gauss = _PackCache()
        self._cache_gauss_skip = _PackCache()

    @property
    def _re(self):
        return getattr(self, self._names[0])

    @property
    def _im(self):
        return getattr(self, self._names[1])

    def _check_supported(self):
        k, s, p, d, g, b, op = self._cfg
        want_pad = (2, 1) if (self._causal and not self._transposed) else (2, 0)
        if not (k == (5, 2) and s == (2, 1) and p == want_pad and d == (1, 1) and g == 1 and b and op == (0, 0)):
            raise NotImplementedError(
                f"{type(self).__name__}: the HIP kernel implements the DCCRN block shape only "
                f"(kernel (5,2), stride (2,1), padding {want_pad}, dilation 1, groups 1, bias); got {self._cfg}")

    def packed(self, fold: Optional[torch.Tensor], cin_used: Optional[int] = None):
        re, im = self._re, self._im
        tensors = (re.weight, im.weight, re.bias, im.bias, fold)
        return self._cache.get(tensors, cin_used, lambda: ops.pack_cconv(
            re.weight.detach(), im.weight.detach(), re.bias.detach(), im.bias.detach(), fold, cin_used, self._transposed))

    def packed_gauss(self, fold: Optional[torch.Tensor], cin_used: Optional[int] = None):
        """(wfrag3, epi, has_fold) of the three-product fp32 kernel (ops.pack_cconv_gauss); fold is applied by its epilogue."""
        re, im = self._re, self._im
        return self._cache_gauss.get((re.weight, im.weight, re.bias, im.bias, fold), cin_used, lambda: ops.pack_cconv_gauss(
            re.weight.detach(), im.weight.detach(), re.bias.detach(), im.bias.detach(), fold, cin_used, self._transposed))

    def gauss_for(self, c0: int, c1: int, fold, cin_used):
        """fp32 mode: the Gauss operands if the three-product kernel serves this launch, else None (-> cgemm_kernel)."""
        if ops.PRECISION != "fp32" or not ops.gauss_supported(c0, c1, self.out_channel):
            return None
        return self.packed_gauss(fold, cin_used)

    def packed_bf16(self, fold: Optional[torch.Tensor], cin_used: Optional[int] = None):
        re, im = self._re, self._im
        return self._cache_bf16.get((re.weight, im.weight, fold), cin_used, lambda: ops.pack_cconv_bf16(
            re.weight.detach(), im.weight.detach(), fold, cin_used, self._transposed))

    def _c1_path(self, c0: int, c1: int, stats=None, skip_div: int = 1) -> bool:
        return (ops.PRECISION == "bf16x3" and self._transposed and self._causal and self.out_channel == 1 and stats is None
                and skip_div == 1 and c0 % 8 == 0 and c1 % 8 == 0)

    def takes_images(self, c0: int, c1: int) -> bool:
        """Eval bf16x3: can this block read split-image sources directly (no conversion)?"""
        return self._c1_path(c0, c1) or (ops.PRECISION == "bf16x3" and self.out_channel % 4 == 0
                                        and ops.bf16_supported(self._transposed, c0, c1, 1, self.out_channel))

    def forward_planar(self, x, *, skip=None, skip_div: int = 1, fold=None, slope=None,
                       stats=None, zero_skip: bool = False, want: str = "planar"):
        """x / skip: Planar or ops.Image.  want: "planar" -> Planar, "image" -> ops.Image, "both" -> (Planar, Image).
        Images are the eval-mode bf16x3 inter-layer format; any other combination converts at the edges."""
        self._check_supported()
        cin_used = x.C if zero_skip else None
        if not zero_skip and x.C + (skip.C if skip is not None else 0) != self.in_channel:
            raise RuntimeError(f"expected {self.in_channel} input channels, got {x.C} + {skip.C if skip is not None else 0}")
        wbf = None
        c1 = skip.C if skip is not None else 0
        any_img = isinstance(x, ops.Image) or isinstance(skip, ops.Image)
        if (ops.PRECISION == "fp32" and not any_img and want == "planar" and skip is not None and skip_div > 1 and stats is None
                and self._transposed and ops.SKIP_ONCE and ops.gauss_supported(x.C, 0, self.out_channel)
                and ops.gauss_supported(c1, 0, self.out_channel)):
            # repeated skips (pvae_module.py:2563-2567: every utterance's skip num_samples times): the conv is linear in its
            # input channels, so the skip half runs ONCE per utterance and is added to each sample's latent half in the
            # epilogue: (ns + 1) / (2 ns) of the multiplications (0.75 at ns = 2, 0.55 at ns = 10), and no strided gather
            re, im = self._re, self._im
            g_skip = self._cache_gauss_skip.get((re.weight, im.weight), x.C, lambda: ops.pack_cconv_gauss_skip_part(
                re.weight.detach(), im.weight.detach(), x.C))
            y_skip = ops.cconv2d(skip, None, None, self.out_channel, transposed=True, causal=self._causal, gauss=g_skip)
            g_main = self.packed_gauss(fold, x.C)
            return ops.cconv2d(x, None, None, self.out_channel, transposed=True, causal=self._causal, slope=slope, gauss=g_main,
                               addend=y_skip, addend_div=skip_div)
        if ops.PRECISION == "fp32" and not any_img and want == "planar":
            g3 = self.gauss_for(x.C, c1, fold, cin_used)
            if g3 is not None:           # fp32: three real products per complex product (csrc/cgemm_gauss.hip)
                return ops.cconv2d(x, None, None, self.out_channel, transposed=self._transposed, causal=self._causal,
                                   slope=slope, skip=skip, skip_div=skip_div, stats=stats, gauss=g3)
        wfrag, bias = self.packed(fold, cin_used)
        if any_img and want == "planar" and self._c1_path(x.C, c1, stats, skip_div):
            # single-output-channel block: image sources, planar result
            x = x if isinstance(x, ops.Image) else ops.to_image(x)
            skip = skip if (skip is None or isinstance(skip, ops.Image)) else ops.to_image(skip)
            re, im = self._re, self._im
            wc1 = self._cache_c1.get((re.weight, im.weight, fold), cin_used, lambda: ops.pack_ctconv_c1(
                re.weight.detach(), im.weight.detach(), fold, cin_used))
            return ops.ctconv_c1(x, wc1, bias, slope=slope, skip=skip)
        if any_img or want != "planar":
            img_ok = (ops.PRECISION == "bf16x3" and stats is None and skip_div == 1 and self.out_channel % 4 == 0
                      and ops.bf16_supported(self._transposed, x.C, c1, 1, self.out_channel))
            if img_ok:
                if any_img:                                  # one source format per launch: lift the planar one
                    x = x if isinstance(x, ops.Image) else ops.to_image(x)
                    skip = skip if (skip is None or isinstance(skip, ops.Image)) else ops.to_image(skip)
                outp, outi = ops.cconv2d_img(x, self.packed_bf16(fold, cin_used), bias, self.out_channel,
                                             transposed=self._transposed, causal=self._causal, slope=slope, skip=skip,
                                             want_planar=want != "image", want_image=want != "planar")
                return outp if want == "planar" else (outi if want == "image" else (outp, outi))
            x = ops.to_planar(x) if isinstance(x, ops.Image) else x
            skip = ops.to_planar(skip) if isinstance(skip, ops.Image) else skip
            if want != "planar" and stats is None and skip_div == 1 and self.out_channel % 4 == 0 and not (
                    ops.PRECISION == "bf16x3" and ops.bf16_supported(self._transposed, x.C, c1, 1, self.out_channel)):
                # exact-fp32 kernel (e.g. the first encoder block, Cin = 1) with an image-writing epilogue
                outp, outi = ops.cconv2d(x, wfrag, bias, self.out_channel, transposed=self._transposed, causal=self._causal,
                                         slope=slope, skip=skip, image="only" if want == "image" else "also")
                return outi if want == "image" else (outp, outi)
            outp = self.forward_planar(x, skip=skip, skip_div=skip_div, fold=fold, slope=slope, stats=stats,
                                       zero_skip=zero_skip)
            return outp if want == "planar" else (ops.to_image(outp) if want == "image" else (outp, ops.to_image(outp)))
        if self._c1_path(x.C, c1, stats, skip_div):
            re, im = self._re, self._im
            wc1 = self._cache_c1.get((re.weight, im.weight, fold), cin_used, lambda: ops.pack_ctconv_c1(
                re.weight.detach(), im.weight.detach(), fold, cin_used))
            return ops.ctconv_c1(x, wc1, bias, slope=slope, skip=skip)
        if ops.PRECISION == "bf16x3" and ops.bf16_supported(self._transposed, x.C, c1, skip_div, self.out_channel):
            wbf = self.packed_bf16(fold, cin_used)
        return ops.cconv2d(x, wfrag, bias, self.out_channel, transposed=self._transposed, causal=self._causal,
                           slope=slope, skip=skip, skip_div=skip_div, stats=stats, wfrag_bf16=wbf)

    def forward(self, x):
        return tag5(self.forward_planar(planar_of(x)))


class ComplexConv2d(_ComplexConvBase):
    """reference: model/complex_progress.py:24-36"""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__(in_channel, out_channel, kernel_size, stride, padding, 0, dilation, groups, bias)


class causal_complex_conv2d(ComplexConv2d):
    """reference: model/complex_progress.py:8-22 (time padding 1, last frame dropped)"""
    _causal = True


class ComplexConvTranspose2d(_ComplexConvBase):
    """reference: model/complex_progress.py:253-279"""
    _transposed = True
    _names = ("tconv_re", "tconv_im")


class causal_ComplexConvTranspose2d(ComplexConvTranspose2d):
    """reference: model/complex_progress.py:222-250"""
    _causal = True


class ComplexBatchNormal(nn.Module):
    """reference: model/complex_progress.py:92-209.  H and W are accepted and unused, as there.
    BN mode follows the explicit ``train`` argument, not ``module.training``."""

    def __init__(self, C, H, W, momentum=0.9, dis_cbn=False):
        super().__init__()
        self.momentum = momentum
        self.gamma_rr = nn.Parameter(torch.ones(C))
        self.gamma_ri = nn.Parameter(torch.randn(C))
        self.gamma_ii = nn.Parameter(torch.ones(C))
        self.beta_r = nn.Parameter(torch.zeros(C))
        self.beta_i = nn.Parameter(torch.zeros(C))
        self.epsilon = 1e-5
        self.register_buffer("running_mean_real", torch.zeros(1, C, 1, 1))
        self.register_buffer("running_mean_imag", torch.zeros(1, C, 1, 1))
        self.register_buffer("Vrr", torch.ones(1, C, 1, 1))
        self.register_buffer("Vri", torch.zeros(1, C, 1, 1))
        self.register_buffer("Vii", torch.ones(1, C, 1, 1))
        self.init_flag = True
        self.dis_cbn = dis_cbn
        self.C = C
        self._cache = _PackCache()
        self._stats_gen = 0         # bumped whenever a kernel rewrites the running buffers through raw pointers

    def _affine(self):
        return [t.detach() for t in (self.gamma_rr, self.gamma_ri, self.gamma_ii, self.beta_r, self.beta_i)]

    def eval_fold(self) -> torch.Tensor:
        """[C, 6] affine of the running statistics (complex_progress.py:161-166 + cbn)."""
        bufs = (self.running_mean_real, self.running_mean_imag, self.Vrr, self.Vri, self.Vii)
        tensors = tuple(bufs) + (self.gamma_rr, self.gamma_ri, self.gamma_ii, self.beta_r, self.beta_i)

        def build():
            mom = torch.stack([b.detach().reshape(-1) for b in bufs]).contiguous()
            return ops.cbn_fold(mom, *self._affine())
        # idv_cbn_finalize updates the running buffers in place behind torch's back (no _version bump), so the
        # generation counter is part of the key; the conv pack caches key on this fold tensor and follow it
        return self._cache.get(tensors, self._stats_gen, build)

    def finish_train(self, act: Planar, stats: torch.Tensor, slope=None):
        """stats (moment sums from the conv epilogue or idv_cbn_stats) -> running buffers, normalise
        `act` in place (+ PReLU when slope is given).  complex_progress.py:131-160."""
        first = bool(self.init_flag)
        if not all(b.is_contiguous() for b in (self.running_mean_real, self.running_mean_imag, self.Vrr, self.Vri, self.Vii)):
            raise RuntimeError("ComplexBatchNormal running buffers must be contiguous")
        ops.cbn_train(act, stats, self, slope, first_call=first, momentum=self.momentum)
        self._stats_gen += 1
        if first and not self.dis_cbn:
            self.init_flag = False
        return act

    def forward(self, x, train=True):
        src = planar_of(x)
        if train and AG.grad_mode(src.buf, self.gamma_rr, self.gamma_ri, self.gamma_ii, self.beta_r, self.beta_i):
            zbuf = AG.BatchNormFn.apply(self, AG._geom(src), src.buf, self.gamma_rr, self.gamma_ri, self.gamma_ii, self.beta_r,
                                        self.beta_i)
            return tag5(ops.rewrap(zbuf, src))
        act = Planar(src.buf.detach().clone(), src.C, src.F, src.B, src.T, src.Tp, src.Jp)   # inputs are not mutated
        if train:
            self.finish_train(act, ops.cbn_stats(act))
        else:
            ops.cbn_apply(act, self.eval_fold())
        return tag5(act)


class ComplexLSTM(nn.Module):
    """reference: model/complex_progress.py:39-74: lstm_re / lstm_im applied to the real and the imaginary
    input, real = rr - ii, imag = ir + ri.  Input [T, B, I, 2] -> [T, B, H, 2]."""

    def __init__(self, input_size, hidden_size, device, num_layers=1, bias=True, dropout=0, bidirectional=False):
        super().__init__()
        self.num_layer = num_layers
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.device = device
        mk = lambda: nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, bias=bias,
                             dropout=dropout, bidirectional=bidirectional)
        self.lstm_re = mk()
        self.lstm_im = mk()
        self._ok = (num_layers == 2 and bias and dropout == 0 and not bidirectional and hidden_size % 16 == 0
                    and input_size % 2 == 0)
        self._cache = _PackCache()

    def _packed(self):
        if not self._ok:
            raise NotImplementedError("ComplexLSTM HIP path: 2 unidirectional layers with bias, hidden_size % 16 == 0, "
                                      "even input_size (the configuration every shipped model uses)")
        params = [q for _, q in sorted(self.named_parameters())]
        sd = dict(self.named_parameters())
        get = lambda n: sd[n].detach()
        H, I = self.hidden_size, self.input_size
        dev = params[0].device
        return self._cache.get(params, None, lambda: (ops.pack_lstm(get, H, I, 0, dev), ops.pack_lstm(get, H, H, 1, dev)))

    def forward_planar(self, x: Planar) -> Planar:
        """x: planar with C*F == input_size feature planes per part -> planar [2][H][Jp]."""
        if x.C * x.F != self.input_size:
            raise RuntimeError(f"ComplexLSTM expects {self.input_size} features, got {x.C}*{x.F}")
        if AG.grad_mode(x.buf, *self.parameters()):
            if not self._ok:
                self._packed()
            return AG.lstm(self, x)
        p0, p1 = self._packed()
        return ops.clstm(x, p0, p1, self.hidden_size)

    def forward(self, x):
        # [T, B, I, 2] -> [B, I, 1, T, 2]
        pl = Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).float())
        out = self.forward_planar(pl)
        y = out.channel_slice(0, self.hidden_size).permute(1, 0, 2, 3)
        y._idv = out
        return y


class ComplexDense(nn.Module):
    """reference: model/complex_progress.py:77-89: two independent real linears (attribute names
    linear_read / linear_imag as in the reference).  Input [N, in, 2] -> [N, out, 2]."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.linear_read = nn.Linear(in_channel, out_channel)
        self.linear_imag = nn.Linear(in_channel, out_channel)
        self.in_channel, self.out_channel = in_channel, out_channel
        self._cache = _PackCache()

    def _packed(self):
        r, im = self.linear_read, self.linear_imag
        return self._cache.get((r.weight, r.bias, im.weight, im.bias), None, lambda: (
            ops.pack_pw(r.weight.detach(), r.bias.detach()) + (ops.pack_pw_bf16(r.weight.detach()),),
            ops.pack_pw(im.weight.detach(), im.bias.detach()) + (ops.pack_pw_bf16(im.weight.detach()),)))

    def forward_planar(self, x: Planar, C_out: int, F_out: int) -> Planar:
        if x.C * x.F != self.in_channel or C_out * F_out != self.out_channel:
            raise RuntimeError("ComplexDense shape mismatch")
        r, im = self.linear_read, self.linear_imag
        if AG.grad_mode(x.buf, r.weight, r.bias, im.weight, im.bias):
            obuf = AG.DenseFn.apply(self, AG._geom(x), (C_out, F_out), x.buf, r.weight, r.bias, im.weight, im.bias)
            return Planar(obuf, C_out, F_out, x.B, x.T, x.Tp, x.Jp)
        pr, pi = self._packed()
        return ops.cdense(x, pr, pi, self.out_channel, C_out, F_out)

    def forward(self, x):
        N = x.shape[0]
        pl = Planar.from_tensor5(x.permute(1, 0, 2).reshape(1, self.in_channel, 1, N, 2).float())
        out = self.forward_planar(pl, self.out_channel, 1)
        return out.channel_slice(0, self.out_channel)[0]
