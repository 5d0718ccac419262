"""torch.autograd integration of the HIP hot path: one ``torch.autograd.Function`` per operator of SURVEY 8(a), whose
``backward`` runs the idv_*_bwd kernels.  This is what makes ``loss.backward(); optimizer.step()`` of the reference's
train steps work on the drop-in modules (supervised_dccrn/train.py:233-243, i_dccrn_vae/pretrained_vaes/train.py:281-301,
i_dccrn_vae/nsvae_dccrn/train_nsvae.py:487-574, train_second_phase_decoder.py:376-433).

The tensors that carry the graph are the flat planar-J buffers (``Planar.buf``); the reference-shaped views handed to the
caller are ordinary strided views of them, so they carry ``grad_fn`` too.  Gradients are planar buffers with the same
geometry and the same invariant (guard columns zero).  There is no CPU path and no torch arithmetic in here: every
``backward`` is a sequence of C-ABI calls.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from . import _lib as L
from ._lib import call, p, i, f, d, ll, stream_ptr
from .ops import Planar


def grad_mode(*tensors) -> bool:
    """Build a graph?  (grad enabled and at least one participating tensor requires grad)"""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _fit(g: Planar, like_buf: torch.Tensor) -> torch.Tensor:
    """Gradient buffer with exactly the numel of the forward buffer (autograd checks shapes)."""
    if g.buf.numel() == like_buf.numel():
        return g.buf
    out = torch.zeros_like(like_buf)
    n = min(out.numel(), g.buf.numel())
    out[:n].copy_(g.buf[:n])
    return out


def _geom(x: Planar):
    return (x.C, x.F, x.B, x.T, x.Tp, x.Jp)


def _mk(buf, geom) -> Planar:
    return Planar(buf, *geom)


# ----------------------------------------------------------------------------- conv block (conv [+ CBN + PReLU])
class ConvBlockFn(torch.autograd.Function):
    """(causal_)ComplexConv2d / ComplexConvTranspose2d on [x | skip], optionally followed by train-mode
    ComplexBatchNormal + PReLU (Encoder / Decoder blocks, pvae_module.py:45-93)."""

    @staticmethod
    def forward(ctx, meta, xbuf, skipbuf, w_re, w_im, b_re, b_im, g_rr, g_ri, g_ii, beta_r, beta_i, slope):
        conv, bn, xg, sg, zero_skip = meta["conv"], meta["bn"], meta["x"], meta["skip"], meta["zero_skip"]
        causal = conv._causal        # non-causal blocks (model/net_config.py: padding (2, 0), T - 1 / T + 1 frames) train as well
        x = _mk(xbuf, xg)
        skip = _mk(skipbuf, sg) if skipbuf is not None else None
        cin_used = x.C if zero_skip else None
        cout = conv.out_channel
        dev = xbuf.device
        div = meta.get("skip_div", 1)
        if div > 1:
            # repeated skips (pvae_module.py:2563-2567), fp32: the skip half of the (linear) transposed conv runs ONCE per
            # utterance and is added to each sample's latent half by the epilogue (conv_block only takes this path when
            # skip_once_ok); backward sums dy over the samples first and runs the skip half's data / weight gradient at batch B
            re, im = conv._re, conv._im
            g_skip = conv._cache_gauss_skip.get((re.weight, im.weight), x.C, lambda: ops.pack_cconv_gauss_skip_part(
                re.weight.detach(), im.weight.detach(), x.C))
            y_skip = ops.cconv2d(skip, None, None, cout, transposed=True, causal=True, gauss=g_skip)
            g_main = conv.packed_gauss(None, x.C)
            stats = torch.zeros(cout, 5, dtype=torch.float64, device=dev) if bn is not None else None
            y = ops.cconv2d(x, None, None, cout, transposed=True, causal=True, stats=stats, gauss=g_main, addend=y_skip,
                            addend_div=div)
            if bn is not None:
                first = bool(bn.init_flag)
                moments, fold = ops.cbn_finalize(stats, float(y.B) * y.F * y.T, bn, first, bn.momentum)
                bn._stats_gen += 1
                if first and not bn.dis_cbn:
                    bn.init_flag = False
                z = ops.cbn_apply_to(y, fold, slope)
                ctx.save_for_backward(xbuf, skipbuf, w_re, w_im, y.buf, fold, moments, g_rr, g_ri, g_ii, slope)
            else:
                z = y
                ctx.save_for_backward(xbuf, skipbuf, w_re, w_im, None, None, None, None, None, None, None)
            ctx.meta = meta
            ctx.zgeom = _geom(z)
            return z.buf
        # bf16x3 training mode: forward and data gradient on the split-bf16 MFMA kernels where the shape allows
        # (every block but the one-channel ends); weight gradients stay on the fp32 MFMA
        wbf = None
        if causal and ops.PRECISION == "bf16x3" and ops.bf16_supported(conv._transposed, x.C, skip.C if skip is not None else 0, 1, cout):
            wbf = conv.packed_bf16(None, cin_used)
        gauss = conv.gauss_for(x.C, skip.C if skip is not None else 0, None, cin_used) if wbf is None else None
        wfrag, bias = conv.packed(None, cin_used) if gauss is None else (None, None)
        if bn is not None:
            stats = torch.zeros(cout, 5, dtype=torch.float64, device=dev)
            if wbf is not None and ops.IMAGE_TRAIN and x.Jp == (skip.Jp if skip is not None else x.Jp):
                # split images of the sources (made once per activation: an encoder output feeds the next block AND, as
                # skip, a decoder block): the image-source kernel stages by LDS-DMA, 1.5x the planar-source one
                y = ops.cconv2d_img_train(_image_of(xbuf, x), wbf, bias, cout, stats, transposed=conv._transposed,
                                          skip=_image_of(skipbuf, skip) if skip is not None else None)
            else:
                y = ops.cconv2d(x, wfrag, bias, cout, transposed=conv._transposed, causal=causal, skip=skip, stats=stats,
                                wfrag_bf16=wbf, gauss=gauss)
            first = bool(bn.init_flag)
            moments, fold = ops.cbn_finalize(stats, float(y.B) * y.F * y.T, bn, first, bn.momentum)
            bn._stats_gen += 1
            if first and not bn.dis_cbn:
                bn.init_flag = False
            if ops.train_image_ok(cout) and y.Jp == x.Jp:
                z, zimg = ops.cbn_apply_to(y, fold, slope, want_image=True)      # the next block reads the image
                z.buf._idv_img, z.buf._idv_img_ver = zimg, z.buf._version
            else:
                z = ops.cbn_apply_to(y, fold, slope)
            ctx.save_for_backward(xbuf, skipbuf, w_re, w_im, y.buf, fold, moments, g_rr, g_ri, g_ii, slope)
        else:
            z = ops.cconv2d(x, wfrag, bias, cout, transposed=conv._transposed, causal=causal, skip=skip, wfrag_bf16=wbf, gauss=gauss)
            ctx.save_for_backward(xbuf, skipbuf, w_re, w_im, None, None, None, None, None, None, None)
        ctx.meta = meta
        ctx.zgeom = _geom(z)
        return z.buf

    @staticmethod
    def backward(ctx, dzbuf):
        meta = ctx.meta
        conv, bn, xg, sg, zero_skip = meta["conv"], meta["bn"], meta["x"], meta["skip"], meta["zero_skip"]
        xbuf, skipbuf, w_re, w_im, ybuf, fold, moments, g_rr, g_ri, g_ii, slope = ctx.saved_tensors
        x = _mk(xbuf, xg)
        skip = _mk(skipbuf, sg) if skipbuf is not None else None
        dz = _mk(dzbuf.contiguous(), ctx.zgeom)
        tr = conv._transposed
        cout = conv.out_channel
        grads_bn = (None,) * 6
        if bn is not None:
            y = _mk(ybuf, ctx.zgeom)
            want_img = _dy_image_wanted(cout) and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
            dy, dgrr, dgri, dgii, dbr, dbi, dslope = ops.cbn_bwd(dz, y, fold, moments, (g_rr, g_ri, g_ii), slope,
                                                                 float(y.B) * y.F * y.T, want_image=want_img)
            if want_img:
                dy, dy_img_bn = dy
            grads_bn = (dgrr, dgri, dgii, dbr, dbi, dslope.reshape(slope.shape) if slope is not None else None)
        else:
            dy = dz
        if bn is None or not want_img:
            dy_img_bn = None
        # bias and weight gradients.  A conv bias in front of a batch norm has an exactly zero gradient: the backward of the
        # normalisation makes every channel of dy sum to zero over the batch (sum dy = Z^T sum du + A sum (y - mu) + N c with
        # sum (y - mu) = 0 and c = -Z^T sum du / N, cbn_bwd_finalize); what autograd returns there in the reference is
        # rounding noise far below the weight-decay term of its Adam step.  No pass over dy for it.
        # frozen layers (requires_grad False on the conv parameters, e.g. train_second_phase_decoder.py:145-149 unfreezes only
        # part of the decoder) skip the weight-gradient contraction, the most expensive backward kernel of a block
        need_w = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        need_b = ctx.needs_input_grad[5] or ctx.needs_input_grad[6]
        db_re = db_im = dw_re = dw_im = None
        if need_b:
            if bn is not None:
                db_re = torch.zeros(cout, dtype=torch.float32, device=dzbuf.device)
                db_im = torch.zeros_like(db_re)
            else:
                db_re, db_im = ops.cconv_bias_grad(dy)
        cin_total = w_re.shape[0] if tr else w_re.shape[1]
        div = meta.get("skip_div", 1)
        dy_sum = None
        if div > 1 and (need_w or (ctx.needs_input_grad[2] and skip is not None)):
            # sum of dy over the num_samples copies of each utterance: what the skip half's gradients contract with
            dy_sum = Planar.empty(dy.C, dy.F, skip.B, dy.T, dy.Tp, dy.buf.device)
            call("idv_repeat_batch_bwd", dy.ptr(), i(div), i(2 * dy.C * dy.F), i(skip.B), i(dy.Tp), i(dy.Jp), i(dy_sum.Jp),
                 dy_sum.ptr(), stream_ptr())
        if need_w:
            used = x.C + (skip.C if skip is not None else 0)
            mk = torch.zeros_like if used < cin_total else torch.empty_like
            dw_re, dw_im = mk(w_re), mk(w_im)
            ops.cconv_wgrad(x, 0, dy, cout, cin_total, tr, conv._causal, dw_re, dw_im)
            if skip is not None:
                ops.cconv_wgrad(skip, x.C, dy_sum if div > 1 else dy, cout, cin_total, tr, conv._causal, dw_re, dw_im)
        # data gradients: the adjoint operator with conjugate-transposed weights
        dx = dskip = None
        need_x, need_s = ctx.needs_input_grad[1], ctx.needs_input_grad[2] and skip is not None
        if need_x or need_s:
            if not tr:      # conv [Cout][Cin]: adjoint = transposed conv, Cin' = Cout, Cout' = Cin (single source)
                if skip is not None:
                    raise NotImplementedError("conv blocks take one source")
                dx = _dgrad(dy, w_re, w_im, cin_total, cout, False, dy_img_bn if dy_img_bn is not None else _dy_image(dy, cout),
                            causal=conv._causal)
            else:           # transposed conv [Cin][Cout]: adjoint = conv, Cout' = a slice of Cin, Cin' = Cout
                per = cout * 10
                dy_img = dy_img_bn if dy_img_bn is not None else _dy_image(dy, cout)
                if need_x:
                    dx = _dgrad(dy, w_re, w_im, x.C, cout, True, dy_img, causal=conv._causal)
                if need_s:
                    wr, wi = w_re.reshape(-1)[x.C * per:], w_im.reshape(-1)[x.C * per:]
                    if div > 1:
                        dskip = _dgrad(dy_sum, wr, wi, skip.C, cout, True, None)
                    else:
                        dskip = _dgrad(dy, wr, wi, skip.C, cout, True, dy_img, causal=conv._causal)
        return (None, _fit(dx, xbuf) if dx is not None else None, _fit(dskip, skipbuf) if dskip is not None else None,
                dw_re, dw_im, db_re, db_im) + grads_bn


def _dgrad(dy: Planar, w_re, w_im, cout_adj: int, cin_adj: int, fwd_transposed: bool, dy_img=None, causal: bool = True) -> Planar:
    """Adjoint operator with conjugate-transposed weights.  bf16x3 mode, where the shape allows: the split-bf16 kernel, fed
    with the split IMAGE of dy (dy_img, made once per block by the caller) when the image kernels take the shape -- they
    stage by LDS-DMA instead of splitting the fp32 patch in registers, 1.5x faster than the planar-source form."""
    adj_tr = not fwd_transposed
    if causal and ops.PRECISION == "bf16x3" and ops.bf16_supported(adj_tr, cin_adj, 0, 1, cout_adj):
        w16 = ops.pack_cconv_bf16_adjoint(w_re, w_im, cout_adj, cin_adj, cin_adj, adj_tr)
        zb = ops.zero_bias(cout_adj, w_re.device)
        if dy_img is not None and cout_adj % 4 == 0:
            return ops.cconv2d_img(dy_img, w16, zb, cout_adj, transposed=adj_tr, causal=True, adjoint=True, want_planar=True,
                                   want_image=False)[0]
        return ops.cconv_dgrad(dy, None, zb, cout_adj, fwd_transposed, True, wfrag_bf16=w16)
    if ops.PRECISION == "fp32" and ops.gauss_supported(cin_adj, 0, cout_adj, bwd=True):
        # exact fp32 with three real products per complex product (csrc/cgemm_gauss.hip)
        g3 = ops.pack_cconv_gauss(w_re, w_im, None, None, None, adjoint_of=(cout_adj, cin_adj, cin_adj, adj_tr))
        return ops.cconv_dgrad(dy, None, None, cout_adj, fwd_transposed, causal, gauss=g3)
    wf, bz = ops.pack_cconv_adjoint(w_re, w_im, cout_adj, cin_adj, cin_adj, adj_tr)
    return ops.cconv_dgrad(dy, wf, bz, cout_adj, fwd_transposed, causal)


def _image_of(buf: torch.Tensor, pl: Planar):
    """Split image of an activation, cached on its buffer tensor (same lifetime as the activation itself; dropped when a
    torch-level in-place op has changed the buffer since)."""
    img = getattr(buf, "_idv_img", None)
    if img is None or getattr(buf, "_idv_img_ver", buf._version) != buf._version or img.Jp != pl.Jp or img.C != pl.C or img.F != pl.F:
        img = ops.to_image(pl)
        buf._idv_img = img
    buf._idv_img_ver = buf._version
    return img


def _dy_image_wanted(cin_adj: int) -> bool:
    return ops.PRECISION == "bf16x3" and ops.IMAGE_TRAIN and cin_adj % 8 == 0 and 2 * cin_adj >= 64


def _dy_image(dy: Planar, cin_adj: int):
    """Split image of dy for the data-gradient kernels of a block (None outside bf16x3 mode / unsupported channel counts)."""
    return ops.to_image(dy) if _dy_image_wanted(cin_adj) else None


def skip_once_ok(conv, x: Planar, skip: Optional[Planar], skip_div: int) -> bool:
    """Can the block run the skip half of its transposed conv once per utterance (fp32, three-product kernel on both halves)?"""
    return (skip is not None and skip_div > 1 and ops.PRECISION == "fp32" and ops.SKIP_ONCE and conv._transposed and conv._causal
            and x.B == skip.B * skip_div and ops.gauss_supported(x.C, 0, conv.out_channel)
            and ops.gauss_supported(skip.C, 0, conv.out_channel))


def conv_block(conv, bn, prelu_weight, x: Planar, skip: Optional[Planar], zero_skip: bool, skip_div: int = 1) -> Planar:
    """skip_div > 1: `skip` holds B utterances, x holds B * skip_div samples (callers check skip_once_ok; otherwise they
    materialise the repeat with repeat_batch and pass skip_div = 1)."""
    re, im = conv._re, conv._im
    conv._check_supported()
    meta = dict(conv=conv, bn=bn, x=_geom(x), skip=_geom(skip) if skip is not None else None, zero_skip=zero_skip,
                skip_div=skip_div)
    bnp = (bn.gamma_rr, bn.gamma_ri, bn.gamma_ii, bn.beta_r, bn.beta_i) if bn is not None else (None,) * 5
    cout = conv.out_channel
    Fout = 2 * x.F - 1 if conv._transposed else (x.F - 1) // 2 + 1
    zbuf = ConvBlockFn.apply(meta, x.buf, skip.buf if skip is not None else None, re.weight, im.weight, re.bias, im.bias, *bnp,
                             prelu_weight if bn is not None else None)
    t_out = x.T if conv._causal else (x.T + 1 if conv._transposed else x.T - 1)
    return Planar(zbuf, cout, Fout, x.B, t_out, x.Tp, x.Jp)


# ----------------------------------------------------------------------------- stand-alone ComplexBatchNormal
class BatchNormFn(torch.autograd.Function):
    """ComplexBatchNormal.forward(x, train=True) on its own (complex_progress.py:127-160)."""

    @staticmethod
    def forward(ctx, bn, geom, xbuf, g_rr, g_ri, g_ii, beta_r, beta_i):
        x = _mk(xbuf, geom)
        stats = ops.cbn_stats(x)
        first = bool(bn.init_flag)
        moments, fold = ops.cbn_finalize(stats, float(x.B) * x.F * x.T, bn, first, bn.momentum)
        bn._stats_gen += 1
        if first and not bn.dis_cbn:
            bn.init_flag = False
        z = ops.cbn_apply_to(x, fold, None)
        ctx.save_for_backward(xbuf, fold, moments, g_rr, g_ri, g_ii)
        ctx.geom = geom
        return z.buf

    @staticmethod
    def backward(ctx, dzbuf):
        xbuf, fold, moments, g_rr, g_ri, g_ii = ctx.saved_tensors
        x = _mk(xbuf, ctx.geom)
        dz = _mk(dzbuf.contiguous(), ctx.geom)
        dy, dgrr, dgri, dgii, dbr, dbi, _ = ops.cbn_bwd(dz, x, fold, moments, (g_rr, g_ri, g_ii), None,
                                                       float(x.B) * x.F * x.T)
        return None, None, _fit(dy, xbuf), dgrr, dgri, dgii, dbr, dbi


# ----------------------------------------------------------------------------- ComplexDense
class DenseFn(torch.autograd.Function):
    """ComplexDense (complex_progress.py:77-89): real linear on the real planes, imag linear on the imag planes."""

    @staticmethod
    def forward(ctx, mod, geom, out_cf, xbuf, w_r, b_r, w_i, b_i):
        x = _mk(xbuf, geom)
        keep = ops.PRECISION
        try:
            ops.PRECISION = "fp32"                      # the training forward is exact fp32
            pr, pi = mod._packed()
            out = ops.cdense(x, pr, pi, mod.out_channel, out_cf[0], out_cf[1])
        finally:
            ops.PRECISION = keep
        ctx.save_for_backward(xbuf, w_r, w_i)
        ctx.geom, ctx.ogeom, ctx.mod = geom, _geom(out), mod
        return out.buf

    @staticmethod
    def backward(ctx, dobuf):
        xbuf, w_r, w_i = ctx.saved_tensors
        x = _mk(xbuf, ctx.geom)
        do = _mk(dobuf.contiguous(), ctx.ogeom)
        M, K = w_r.shape
        J = x.B * x.Tp
        dev = xbuf.device
        dws = [torch.empty_like(w_r), torch.empty_like(w_i)]
        dbs = [torch.empty(M, dtype=torch.float32, device=dev) for _ in range(2)]
        dx = ops.like(x) if ctx.needs_input_grad[3] else None
        for ri, w in enumerate((w_r, w_i)):
            ops.pw_wgrad(do.ptr(ri * do.C), M, do.Jp, x.ptr(ri * x.C), K, x.Jp, J, dws[ri])
            ops.planar_rowsum(do.ptr(ri * do.C), M, do.Jp, J, dbs[ri])
            if dx is not None:
                wT = ops.pack_pw(w.t().contiguous(), None)
                ops.pw_gemm(do.ptr(ri * do.C), M, wT[0], wT[1], K, x.B, x.Tp, do.Jp, x.T, dx.ptr(ri * x.C))
        return None, None, None, _fit(dx, xbuf) if dx is not None else None, dws[0], dbs[0], dws[1], dbs[1]


# ----------------------------------------------------------------------------- ComplexLSTM
_LSTM_NAMES = [f"{m}.{n}_l{l}" for m in ("lstm_re", "lstm_im") for l in (0, 1)
               for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


def _colp_perm(H: int, device) -> torch.Tensor:
    """perm[colp] = torch gate row (gate*H + unit) for the recurrent kernels' gate-column order."""
    colp = torch.arange(4 * H, device=device)
    ub, g, ul = colp >> 6, (colp >> 4) & 3, colp & 15
    return g * H + ub * 16 + ul


class LstmFn(torch.autograd.Function):
    """ComplexLSTM.forward (complex_progress.py:50-74) with back-propagation through time."""

    @staticmethod
    def forward(ctx, mod, geom, xbuf, *params):
        x = _mk(xbuf, geom)
        H, K = mod.hidden_size, mod.input_size
        p0, p1 = mod._packed()
        out = Planar.empty(H, 1, x.B, x.T, x.Tp, xbuf.device)
        nwork = ops._ll_fn("idv_clstm_train_work_floats")(i(H), i(x.B), i(x.T), i(x.Jp))
        work = torch.empty(int(nwork), dtype=torch.float32, device=xbuf.device)
        # bf16x3 training mode: the split-bf16 recurrences (H = 128 register-resident, H = 384 / 768 persistent cooperative)
        # also save the gates and cell states the fp32 BPTT kernels read; every other size keeps the exact-fp32 recurrence
        flags = 4 if ops.LSTM_PERSISTENT else 4 | 8            # bit 3: per-step / one-CU kernels instead of the cooperative ones
        if ops.PRECISION == "bf16x3" and (H == 128 or (ops.LSTM_PERSISTENT and L.lib().idv_lstm_pers_supported(i(H), i(x.B)))):
            flags |= 1
        wih1_16 = None
        if flags & 1:
            # input projections on the bf16 MFMA as in eval: layer 0 from the K-major split image of x, layer 1 from the
            # image of h0 the persistent recurrence writes (H = 384 / 768)
            if p0[3] is not None:
                kimg = ops.KImage.from_planes(x.ptr(), 2 * K, x.B * x.Tp, x.Jp, xbuf.device)
                call("idv_lstm_proj_bf16x3", kimg.ptr(), ll(kimg.lo_slots), i(K), p(p0[3]), p(p0[1]), p(work), i(H), i(x.B), i(x.T),
                     i(x.Tp), i(x.Jp), stream_ptr())
                flags |= 2
            wih1_16 = p1[3]
        call("idv_clstm_fwd2", x.ptr(), i(K), p(p0[0]), p(p0[1]), p(p0[2]), p(p1[0]), p(p1[1]), p(p1[2]),
             p(p1[4] if (ops.LSTM_STACK2 and len(p1) > 4) else None), i(H), i(x.B), i(x.T), i(x.Tp), i(x.Jp), p(work), out.ptr(),
             i(flags), p(wih1_16), stream_ptr())
        ctx.save_for_backward(xbuf, work, *params)
        ctx.mod, ctx.geom, ctx.ogeom = mod, geom, _geom(out)
        return out.buf

    @staticmethod
    def backward(ctx, dobuf):
        xbuf, work, *params = ctx.saved_tensors
        # idv_lstm_bptt replaces the saved gate activations in `work` by gate gradients through raw pointers (torch's
        # saved-tensor version check cannot see that): a second backward over the same graph would silently be wrong
        if getattr(ctx, "_idv_consumed", False):
            raise RuntimeError("ComplexLSTM backward ran twice over one forward (retain_graph=True / two losses sharing the "
                               "LSTM): the saved gate buffer is consumed in place by the first pass; run forward again")
        ctx._idv_consumed = True
        P = dict(zip(_LSTM_NAMES, params))
        mod = ctx.mod
        x = _mk(xbuf, ctx.geom)
        H, K, B, T, Tp, Jp = mod.hidden_size, mod.input_size, x.B, x.T, x.Tp, x.Jp
        dev = xbuf.device
        J = B * Tp
        TBH = T * B * H
        G0, G1 = work[0:16 * TBH], work[16 * TBH:32 * TBH]
        h0, h1 = work[32 * TBH:36 * TBH], work[36 * TBH:40 * TBH]
        c0, c1 = work[40 * TBH:44 * TBH], work[44 * TBH:48 * TBH]
        do = _mk(dobuf.contiguous(), ctx.ogeom)
        st = stream_ptr
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        bw = new(int(ops._ll_fn("idv_lstm_bptt_work_floats")(i(H), i(B))))
        perm = _colp_perm(H, dev)

        def whhT(l):
            t = new(2 * 4 * H * H)
            call("idv_pack_lstm_hh_bwd", p(P[f"lstm_re.weight_hh_l{l}"]), p(P[f"lstm_im.weight_hh_l{l}"]), i(H), p(t), st())
            return t

        def planar_of_rows(src, ld, c0_, ncol):
            dst = new(ncol * Jp)
            call("idv_rows_to_planar", p(src), ll(ld), i(c0_), i(ncol), i(B), i(T), i(Tp), i(Jp), p(dst), st())
            return dst

        # ---- layer 1
        dh1 = new(4 * TBH)
        call("idv_lstm_uncombine", do.ptr(), i(H), i(B), i(T), i(Tp), i(Jp), p(dh1), st())
        # H = 128: the BPTT of BOTH layers in one cooperative launch, layer 0 a step behind layer 1 and dh0 = dG1 W_ih1 inside
        # the recurrence (csrc/lstm_bptt_stack2_f32.hip; exact fp32 in either arithmetic mode)
        fused = ops.LSTM_STACK2 and ops.LSTM_PERSISTENT and bool(L.lib().idv_lstm_bptt_stack2_supported(i(H), i(B)))
        if fused:
            wihT1 = new(2 * 4 * H * H)
            call("idv_pack_lstm_hh_bwd", p(P["lstm_re.weight_ih_l1"]), p(P["lstm_im.weight_ih_l1"]), i(H), p(wihT1), st())
            bw2 = new((int(ops._ll_fn("idv_lstm_bptt_stack2_work_bytes")(i(H), i(B))) + 3) // 4)
            call("idv_lstm_bptt_stack2", p(G1), p(G0), ll(T * B * 8 * H), ll(4 * H), i(8 * H), p(c1), p(c0), p(dh1), p(whhT(1)),
                 p(wihT1), p(whhT(0)), i(H), i(B), i(T), p(bw2), st())
        else:
            call("idv_lstm_bptt", p(G1), ll(2 * T * B * 4 * H), ll(T * B * 4 * H), i(4 * H), p(c1), p(dh1), p(whhT(1)), i(H), i(B),
                 i(T), p(bw), st())
        h0p = [planar_of_rows(h0[r * TBH:(r + 1) * TBH], H, 0, H) for r in range(4)]
        h1p = [planar_of_rows(h1[r * TBH:(r + 1) * TBH], H, 0, H) for r in range(4)]
        dwih1, dwhh1, db1 = new(8 * H, H), new(8 * H, H), new(8 * H)
        dh0 = None if fused else new(4 * TBH)
        # bf16x3 training mode: the two data-gradient contractions of the projections (dh0 = dG1 W_ih1, dx = dG0 W_ih0) on the
        # split-bf16 point-wise kernel, fed with K-major split images of the gate gradients (the fp32 form was 17 % of the
        # NSVAE train step)
        bf16 = ops.PRECISION == "bf16x3" and (4 * H) % 64 == 0 and H >= 64
        zb = torch.zeros(max(H, K), dtype=torch.float32, device=dev) if bf16 else None
        for s in range(2):
            if not fused:
                w_colp_T = P[f"{'lstm_im' if s else 'lstm_re'}.weight_ih_l1"][perm].t().contiguous()     # [H][4H colp]
                wT = ops.pack_pw_bf16(w_colp_T) if bf16 else ops.pack_pw(w_colp_T, None)
            for k, run in enumerate((s, 2 + s)):
                dG1p = planar_of_rows(G1[run * 4 * TBH:(run + 1) * 4 * TBH], 4 * H, 0, 4 * H)
                acc = k > 0
                ops.pw_wgrad(p(dG1p), 4 * H, Jp, p(h1p[run]), H, Jp, J, dwhh1[s * 4 * H:(s + 1) * 4 * H], shift=-1, rowmap=1, H=H,
                             accumulate=acc)
                ops.pw_wgrad(p(dG1p), 4 * H, Jp, p(h0p[run]), H, Jp, J, dwih1[s * 4 * H:(s + 1) * 4 * H], rowmap=1, H=H,
                             accumulate=acc)
                call("idv_lstm_bias_grad", p(dG1p), i(H), i(Jp), i(J), i(1 if acc else 0), p(db1[s * 4 * H:(s + 1) * 4 * H]), st())
                # gradient arriving at layer 0's output: dG1 W_ih1, written row-major [T*B][H]
                if fused:
                    pass
                elif bf16:
                    kimg = ops.KImage.from_planes(p(dG1p), 4 * H, J, Jp, dev, pad_to=64)
                    ops.pw_bf16x3_rows(kimg, 4 * H, wT, zb, H, H, B, T, Tp, p(dh0[run * TBH:(run + 1) * TBH]))
                else:
                    ops.pw_gemm(p(dG1p), 4 * H, wT[0], wT[1], H, B, Tp, Jp, T, p(dh0[run * TBH:(run + 1) * TBH]), swap=True, ldo=H)
        # ---- layer 0
        if not fused:
            call("idv_lstm_bptt", p(G0), ll(T * B * 8 * H), ll(4 * H), i(8 * H), p(c0), p(dh0), p(whhT(0)), i(H), i(B), i(T),
                 p(bw), st())
        dwih0, dwhh0, db0 = new(8 * H, K), new(8 * H, H), new(8 * H)
        dx = ops.like(x) if ctx.needs_input_grad[2] else None
        if dx is not None:
            wcat = torch.cat([P["lstm_re.weight_ih_l0"][perm], P["lstm_im.weight_ih_l0"][perm]], 0)   # [8H colp][K]
            wT0 = ops.pack_pw_bf16(wcat.t().contiguous()) if bf16 else ops.pack_pw(wcat.t().contiguous(), None)
        for z in range(2):
            dG0p = planar_of_rows(G0[z * 8 * TBH:(z + 1) * 8 * TBH], 8 * H, 0, 8 * H)                # [s][4H colp] planes
            ops.pw_wgrad(p(dG0p), 8 * H, Jp, x.ptr(z * x.C), K, Jp, J, dwih0, rowmap=1, H=H, accumulate=z > 0)
            for s in range(2):
                sl = dG0p[s * 4 * H * Jp:(s + 1) * 4 * H * Jp]
                ops.pw_wgrad(p(sl), 4 * H, Jp, p(h0p[2 * z + s]), H, Jp, J, dwhh0[s * 4 * H:(s + 1) * 4 * H], shift=-1, rowmap=1,
                             H=H, accumulate=z > 0)
                call("idv_lstm_bias_grad", p(sl), i(H), i(Jp), i(J), i(1 if z > 0 else 0), p(db0[s * 4 * H:(s + 1) * 4 * H]), st())
            if dx is not None and bf16:
                kimg = ops.KImage.from_planes(p(dG0p), 8 * H, J, Jp, dev, pad_to=64)
                ops.pw_bf16x3(kimg, 0, 8 * H, wT0, None, K, B, Tp, T, dx.ptr(z * x.C))
            elif dx is not None:
                ops.pw_gemm(p(dG0p), 8 * H, wT0[0], wT0[1], K, B, Tp, Jp, T, dx.ptr(z * x.C))
        grads = {}
        for s, m in enumerate(("lstm_re", "lstm_im")):
            rows = slice(s * 4 * H, (s + 1) * 4 * H)
            grads[f"{m}.weight_ih_l0"], grads[f"{m}.weight_hh_l0"] = dwih0[rows], dwhh0[rows]
            grads[f"{m}.bias_ih_l0"] = grads[f"{m}.bias_hh_l0"] = db0[rows]
            grads[f"{m}.weight_ih_l1"], grads[f"{m}.weight_hh_l1"] = dwih1[rows], dwhh1[rows]
            grads[f"{m}.bias_ih_l1"] = grads[f"{m}.bias_hh_l1"] = db1[rows]
        return (None, None, _fit(dx, xbuf) if dx is not None else None) + tuple(grads[n] for n in _LSTM_NAMES)


def lstm(mod, x: Planar) -> Planar:
    sd = dict(mod.named_parameters())
    obuf = LstmFn.apply(mod, _geom(x), x.buf, *[sd[n] for n in _LSTM_NAMES])
    return Planar(obuf, mod.hidden_size, 1, x.B, x.T, x.Tp, x.Jp)


# ----------------------------------------------------------------------------- STFT / ISTFT / mask
class StftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, signal):
        keep = ops.PRECISION
        try:
            ops.PRECISION = "fp32"
            X = ops.stft(signal, plan)
        finally:
            ops.PRECISION = keep
        ctx.plan, ctx.geom, ctx.L = plan, _geom(X), signal.shape[1]
        return X.buf

    @staticmethod
    def backward(ctx, dXbuf):
        return None, ops.stft_bwd(_mk(dXbuf.contiguous(), ctx.geom), ctx.plan, ctx.L)


class IstftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, geom, specbuf):
        keep = ops.PRECISION
        try:
            ops.PRECISION = "fp32"
            y = ops.istft(_mk(specbuf, geom), plan)
        finally:
            ops.PRECISION = keep
        ctx.plan, ctx.geom = plan, geom
        ctx.save_for_backward(specbuf)
        return y

    @staticmethod
    def backward(ctx, dy):
        (specbuf,) = ctx.saved_tensors
        g = ops.istft_bwd(dy.float(), _mk(specbuf, ctx.geom), ctx.plan)
        return None, None, _fit(g, specbuf)


class MaskFn(torch.autograd.Function):
    """Mask branch of DCCRN_.forward / the fine-tuned decoder (pvae_module.py:224-234, :2594-2608)."""

    @staticmethod
    def forward(ctx, geom, xgeom, x_div, maskbuf, Xbuf):
        mask, X = _mk(maskbuf, geom), _mk(Xbuf, xgeom)
        pred, pc = ops.mask_apply(mask, X, x_div)
        ctx.save_for_backward(maskbuf, Xbuf)
        ctx.geom, ctx.xgeom, ctx.x_div = geom, xgeom, x_div
        ctx.set_materialize_grads(False)
        return pred.buf, torch.view_as_real(pc)

    @staticmethod
    def backward(ctx, dpredbuf, dpc):
        maskbuf, Xbuf = ctx.saved_tensors
        if dpredbuf is None and dpc is None:
            return None, None, None, None, None
        mask, X = _mk(maskbuf, ctx.geom), _mk(Xbuf, ctx.xgeom)
        dpred = _mk(dpredbuf.contiguous(), ctx.geom) if dpredbuf is not None else None
        want_dx = ctx.needs_input_grad[4]
        if want_dx and ctx.x_div != 1:
            raise NotImplementedError("gradient w.r.t. an input spectrum shared by several samples")
        dm, dX = ops.mask_apply_bwd(mask, X, ctx.x_div, dpred, dpc.contiguous() if dpc is not None else None, want_dx)
        return None, None, None, _fit(dm, maskbuf), _fit(dX, Xbuf) if dX is not None else None


class DatanormFn(torch.autograd.Function):
    """Input normalisation of DCCRN_.forward (pvae_module.py:217-221)."""

    @staticmethod
    def forward(ctx, geom, mean, std, Xbuf):
        X = _mk(Xbuf, geom)
        out = Planar.empty(1, X.F, X.B, X.T, X.Tp, Xbuf.device)
        call("idv_datanorm", X.ptr(), p(mean), p(std), i(X.F), i(X.B), i(X.T), i(X.Tp), i(X.Jp), out.ptr(), stream_ptr())
        ctx.geom = geom
        ctx.save_for_backward(std, Xbuf)
        return out.buf

    @staticmethod
    def backward(ctx, dobuf):
        std, Xbuf = ctx.saved_tensors
        X = _mk(Xbuf, ctx.geom)
        do = _mk(dobuf.contiguous(), ctx.geom)
        dX = ops.like(X)
        call("idv_datanorm_bwd", do.ptr(), p(std), i(X.F), i(X.B), i(X.T), i(X.Tp), i(X.Jp), dX.ptr(), stream_ptr())
        return None, None, None, _fit(dX, Xbuf)


class DatadenormFn(torch.autograd.Function):
    """predict = data_std * predict + data_mean (pvae_module.py:235-238, :246-247) -> (planar, interleaved complex)."""

    @staticmethod
    def forward(ctx, geom, mean, std, Pbuf):
        P_ = _mk(Pbuf, geom)
        out = Planar.empty(1, P_.F, P_.B, P_.T, P_.Tp, Pbuf.device)
        pc = torch.empty(P_.B, P_.F, P_.T, 2, dtype=torch.float32, device=Pbuf.device)
        call("idv_datadenorm", P_.ptr(), p(mean), p(std), i(P_.F), i(P_.B), i(P_.T), i(P_.Tp), i(P_.Jp), out.ptr(), p(pc), stream_ptr())
        ctx.geom = geom
        ctx.save_for_backward(std, Pbuf)
        ctx.set_materialize_grads(False)
        return out.buf, pc

    @staticmethod
    def backward(ctx, dobuf, dpc):
        std, Pbuf = ctx.saved_tensors
        if dobuf is None and dpc is None:
            return None, None, None, None
        P_ = _mk(Pbuf, ctx.geom)
        do = _mk(dobuf.contiguous(), ctx.geom) if dobuf is not None else None
        dP = ops.like(P_)
        call("idv_datadenorm_bwd", do.ptr() if do is not None else p(None), p(dpc.contiguous().float()) if dpc is not None else p(None),
             p(std), i(P_.F), i(P_.B), i(P_.T), i(P_.Tp), i(P_.Jp), dP.ptr(), stream_ptr())
        return None, None, None, _fit(dP, Pbuf)


class PlanarToComplexFn(torch.autograd.Function):
    """recon_type 'real_imag' (pvae_module.py:245-253): planar [2][1][F][Jp] -> interleaved [B, F, T, 2]."""

    @staticmethod
    def forward(ctx, geom, buf):
        ctx.geom = geom
        ctx.save_for_backward(buf)
        return torch.view_as_real(ops.planar_to_complex(_mk(buf, geom)))

    @staticmethod
    def backward(ctx, dpc):
        (buf,) = ctx.saved_tensors
        g = ops.complex_to_planar(dpc.contiguous().float(), ctx.geom[4])
        return None, _fit(g, buf)


class RepeatBatchFn(torch.autograd.Function):
    """skip.repeat over num_samples (pvae_module.py:2563-2567) as a materialised planar buffer; backward sums the copies."""

    @staticmethod
    def forward(ctx, geom, n, buf):
        out = ops.repeat_batch(_mk(buf, geom), n)
        ctx.geom, ctx.n, ctx.ogeom = geom, n, _geom(out)
        ctx.save_for_backward(buf)
        return out.buf

    @staticmethod
    def backward(ctx, dobuf):
        (buf,) = ctx.saved_tensors
        x = _mk(buf, ctx.geom)
        do = _mk(dobuf.contiguous(), ctx.ogeom)
        dx = ops.like(x)
        call("idv_repeat_batch_bwd", do.ptr(), i(ctx.n), i(2 * x.C * x.F), i(x.B), i(x.Tp), i(do.Jp), i(x.Jp), dx.ptr(),
             stream_ptr())
        return None, None, _fit(dx, buf)


def repeat_batch(x: Planar, n: int) -> Planar:
    if grad_mode(x.buf):
        obuf = RepeatBatchFn.apply(_geom(x), n, x.buf)
        return Planar(obuf, x.C, x.F, x.B * n, x.T, x.Tp, Planar.jp_for(x.B * n, x.Tp))
    return ops.repeat_batch(x, n)


# ----------------------------------------------------------------------------- reparameterisation
class ReparamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geom, off, zdim, ns, latbuf, eps_r, eps_i):
        lat = _mk(latbuf, geom)
        eps_r, eps_i = eps_r.contiguous().float(), eps_i.contiguous().float()
        z = ops.reparam(lat, off, zdim, eps_r, eps_i, ns)
        ctx.save_for_backward(latbuf, eps_r, eps_i)
        ctx.geom, ctx.off, ctx.zdim, ctx.ns, ctx.zgeom = geom, off, zdim, ns, _geom(z)
        return z.buf

    @staticmethod
    def backward(ctx, dzbuf):
        latbuf, eps_r, eps_i = ctx.saved_tensors
        lat = _mk(latbuf, ctx.geom)
        dlat = _mk(torch.zeros_like(latbuf), ctx.geom)
        ops.reparam_bwd(lat, ctx.off, ctx.zdim, eps_r, eps_i, ctx.ns, _mk(dzbuf.contiguous(), ctx.zgeom), dlat)
        return None, None, None, None, dlat.buf, None, None


# ----------------------------------------------------------------------------- losses
class SisnrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, source, est, src_div):
        B, Ln = est.shape
        work = torch.empty(3 * B, dtype=torch.float64, device=est.device)
        out = torch.empty(1, dtype=torch.float32, device=est.device)
        call("idv_sisnr", p(source), i(source.stride(0)), i(src_div), p(est), i(est.stride(0)), i(B), i(Ln), p(work), p(out),
             stream_ptr())
        ctx.save_for_backward(source, est, work)
        ctx.src_div = src_div
        return out[0]

    @staticmethod
    def backward(ctx, g):
        source, est, work = ctx.saved_tensors
        B, Ln = est.shape
        dest = torch.empty(B, Ln, dtype=torch.float32, device=est.device)
        call("idv_sisnr_bwd", p(source), i(source.stride(0)), i(ctx.src_div), p(est), i(est.stride(0)), i(B), i(Ln), p(work),
             p(g.contiguous().float()), p(dest), stream_ptr())
        return None, dest, None


class ReconLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pr, ori, ori_div):
        B, F, T, _ = pr.shape
        work = torch.empty(3, dtype=torch.float64, device=pr.device)
        out = torch.empty(2, dtype=torch.float32, device=pr.device)
        sb, sf, st, sr = ori.stride()
        call("idv_recon_loss", p(pr), p(ori), ll(sb), ll(sf), ll(st), ll(sr), i(ori_div), i(B), i(F), i(T), p(work), p(out),
             stream_ptr())
        ctx.save_for_backward(pr, ori)
        ctx.ori_div = ori_div
        ctx.set_materialize_grads(False)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cpx, g_mag):
        pr, ori = ctx.saved_tensors
        if g_cpx is None and g_mag is None:
            return None, None, None
        B, F, T, _ = pr.shape
        sb, sf, st, sr = ori.stride()
        dpr = torch.empty_like(pr)
        call("idv_recon_loss_bwd", p(pr), p(ori), ll(sb), ll(sf), ll(st), ll(sr), i(ctx.ori_div), i(B), i(F), i(T),
             p(g_cpx.contiguous().float() if g_cpx is not None else None),
             p(g_mag.contiguous().float() if g_mag is not None else None), p(dpr), stream_ptr())
        return dpr, None, None


class CklFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g1, off1, g2, off2, zdim, eps, q1buf, q2buf):
        q1 = _mk(q1buf, g1)
        q2 = _mk(q2buf, g2) if q2buf is not None else None
        if ctx.needs_input_grad[7]:
            raise NotImplementedError("KL gradient w.r.t. the second (target) posterior: it is frozen in every shipped recipe")
        out = ops.ckl(q1, off1, q2, off2, zdim, eps)
        ctx.save_for_backward(q1buf, q2buf)
        ctx.args = (g1, off1, g2, off2, zdim, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        q1buf, q2buf = ctx.saved_tensors
        g1, off1, g2, off2, zdim, eps = ctx.args
        q1 = _mk(q1buf, g1)
        q2 = _mk(q2buf, g2) if q2buf is not None else None
        dq1 = torch.zeros_like(q1buf)
        o2 = off2 if q2 is not None else (0, 0, 0)
        call("idv_ckl_bwd", q1.ptr(), i(q1.C), i(q1.Jp), i(off1[0]), i(off1[1]), i(off1[2]),
             q2.ptr() if q2 is not None else p(None), i(q2.C if q2 is not None else 0), i(q2.Jp if q2 is not None else 0),
             i(o2[0]), i(o2[1]), i(o2[2]), i(zdim), f(eps), i(q1.B), i(q1.T), i(q1.Tp), p(g.contiguous().float()),
             _mk(dq1, g1).ptr(), stream_ptr())
        return None, None, None, None, None, None, dq1, None


class MiFn(torch.autograd.Function):
    """mutual_information (pretrain_pvaes_loss.py:129-159): gradient to the posterior (miu, log_sigma, delta) and to the samples."""

    @staticmethod
    def forward(ctx, glat, off, gz, zdim, ns, eps, latbuf, zbuf):
        out, work = ops.mi_estimate(_mk(latbuf, glat), off, _mk(zbuf, gz), zdim, ns, eps)
        ctx.save_for_backward(latbuf, zbuf, work)
        ctx.args = (glat, off, gz, zdim, ns, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        latbuf, zbuf, work = ctx.saved_tensors
        glat, off, gz, zdim, ns, eps = ctx.args
        dlat = torch.zeros_like(latbuf) if ctx.needs_input_grad[6] else None
        dz = torch.zeros_like(zbuf) if ctx.needs_input_grad[7] else None
        ops.mi_bwd(_mk(latbuf, glat), off, _mk(zbuf, gz), zdim, ns, eps, work, g.contiguous().float(),
                   _mk(dlat, glat) if dlat is not None else None, _mk(dz, gz) if dz is not None else None)
        return None, None, None, None, None, None, dlat, dz


class MiuDistFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g1, off1, g2, off2, zdim, q1buf, q2buf):
        out = ops.miu_dist(_mk(q1buf, g1), off1, _mk(q2buf, g2), off2, zdim)
        ctx.save_for_backward(q1buf, q2buf, out.reshape(1).clone())
        ctx.args = (g1, off1, g2, off2, zdim)
        return out

    @staticmethod
    def backward(ctx, g):
        q1buf, q2buf, val = ctx.saved_tensors
        g1, off1, g2, off2, zdim = ctx.args
        q1, q2 = _mk(q1buf, g1), _mk(q2buf, g2)
        d1 = torch.zeros_like(q1buf) if ctx.needs_input_grad[5] else None
        d2 = torch.zeros_like(q2buf) if ctx.needs_input_grad[6] else None
        call("idv_miu_dist_bwd", q1.ptr(), i(q1.C), i(q1.Jp), i(off1), q2.ptr(), i(q2.C), i(q2.Jp), i(off2), i(zdim), i(q1.B),
             i(q1.T), i(q1.Tp), p(g.contiguous().float()), p(val), _mk(d1, g1).ptr() if d1 is not None else p(None),
             _mk(d2, g2).ptr() if d2 is not None else p(None), stream_ptr())
        return None, None, None, None, None, d1, d2
