"""Thin host wrappers over the C ABI: planar-J activation buffers and one python function per
idv_* operator.  PyTorch is used for device memory and streams only; all arithmetic of the hot
path happens inside libidccrn_hip.so."""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch

from . import _lib as L
from ._lib import call, p, i, f, d, ll, stream_ptr

SLACK = 256

# Arithmetic of the complex conv / transposed-conv contraction:
#   "fp32"   exact fp32 MFMA (v_mfma_f32_32x32x2_f32)                       -- default
#   "bf16x3" split-precision bf16 MFMA, fp32 accumulate (3 MFMAs per k block) -- ~1e-5 waveform error, ~2x faster
PRECISION = os.environ.get("IDV_PRECISION", "fp32")


def set_precision(mode: str):
    global PRECISION
    if mode not in ("fp32", "bf16x3"):
        raise ValueError("precision must be 'fp32' or 'bf16x3'")
    PRECISION = mode


def bucket(n: int) -> int:
    """Round a buffer length up to one of 8 size classes per octave (<= 12.5 % slack).  Layer outputs of
    slightly different sizes then land in the same class, so the caching allocator re-uses whole blocks
    instead of splitting multi-GB ones and calling hipMalloc again on the next step."""
    if n <= 4096:
        return n
    step = 1 << (n.bit_length() - 4)
    return (n + step - 1) // step * step


class Planar:
    """A planar-J activation  act[2][C][F][Jp]  (see include/idccrn_hip.h).

    ``T`` is the number of valid frames (t_valid); ``Tp`` = T_stft + 1 columns per utterance.
    ``tensor5()`` exposes it with the reference's shape [B, C, F, T, 2] as a strided view.
    """

    __slots__ = ("buf", "C", "F", "B", "T", "Tp", "Jp")

    def __init__(self, buf, C, F, B, T, Tp, Jp):
        self.buf, self.C, self.F, self.B, self.T, self.Tp, self.Jp = buf, C, F, B, T, Tp, Jp

    @staticmethod
    def jp_for(B: int, Tp: int) -> int:
        return (B * Tp + 3) // 4 * 4

    @classmethod
    def empty(cls, C, F, B, T, Tp, device, zero=False):
        Jp = cls.jp_for(B, Tp)
        n = bucket(2 * C * F * Jp + 2 * SLACK)
        buf = (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=device)
        return cls(buf, C, F, B, T, Tp, Jp)

    @property
    def data(self) -> torch.Tensor:
        return self.buf[SLACK:SLACK + 2 * self.C * self.F * self.Jp]

    def ptr(self, plane_offset: int = 0):
        """Device pointer to plane `plane_offset` (planes are F*Jp floats... of size Jp per row)."""
        return L._P(self.buf.data_ptr() + 4 * (SLACK + plane_offset * self.F * self.Jp))

    def planes(self) -> torch.Tensor:
        """[2, C, F, B, Tp] view (requires Jp == B*Tp padding handled by as_strided)."""
        return torch.as_strided(self.buf, (2, self.C, self.F, self.B, self.Tp),
                                (self.C * self.F * self.Jp, self.F * self.Jp, self.Jp, self.Tp, 1), SLACK)

    def tensor5(self) -> torch.Tensor:
        """Reference-shaped view [B, C, F, T, 2]."""
        return self.planes()[..., 1:1 + self.T].permute(3, 1, 2, 4, 0)

    def tensor4(self) -> torch.Tensor:
        """[B, F, T, 2] view for C == 1 (or [B, C*F...] callers reshape themselves)."""
        return self.tensor5()[:, 0]

    def channel_slice(self, c0: int, c1: int) -> torch.Tensor:
        """[B, T, c1-c0, 2] view for F == 1 activations (LSTM outputs / latents)."""
        assert self.F == 1
        return self.planes()[:, c0:c1, 0, :, 1:1 + self.T].permute(2, 3, 1, 0)

    @classmethod
    def from_tensor5(cls, x: torch.Tensor, Tp: Optional[int] = None):
        """Copy a reference-layout tensor [B, C, F, T, 2] into a fresh planar buffer."""
        B, C, F, T, _ = x.shape
        Tp = Tp or T + 1
        out = cls.empty(C, F, B, T, Tp, x.device, zero=True)
        out.tensor5().copy_(x)
        return out


IMG_SLACK = 4096          # bf16 elements kept readable before, between and after the two planes of an Image
IMAGE_PATH = os.environ.get("IDV_IMAGE_PATH", "1") != "0"     # eval bf16x3: keep inter-layer activations as split images
IMAGE_TRAIN = os.environ.get("IDV_IMAGE_TRAIN", "1") != "0"   # bf16x3 training: feed the conv kernels split images too


class Image:
    """A split-bf16 activation image  img[hi|lo][(2C+7)//8][F][Jp][8]  (include/idccrn_hip.h, "split image"): what the
    bf16x3 conv kernels stage into LDS verbatim.  Same (C, F, B, T, Tp, Jp) meaning as Planar."""

    __slots__ = ("buf", "C", "F", "B", "T", "Tp", "Jp", "lo_off")

    def __init__(self, buf, C, F, B, T, Tp, Jp, lo_off):
        self.buf, self.C, self.F, self.B, self.T, self.Tp, self.Jp, self.lo_off = buf, C, F, B, T, Tp, Jp, lo_off

    @classmethod
    def empty(cls, C, F, B, T, Tp, device):
        Jp = Planar.jp_for(B, Tp)
        plane = (2 * C + 7) // 8 * F * Jp * 8
        lo_off = plane + IMG_SLACK
        buf = torch.empty(bucket(2 * IMG_SLACK + lo_off + plane), dtype=torch.int16, device=device)
        return cls(buf, C, F, B, T, Tp, Jp, lo_off)

    def ptr(self):
        return L._P(self.buf.data_ptr() + 2 * IMG_SLACK)

    @property
    def lo_slots(self) -> int:
        return self.lo_off // 8


def to_image(x: Planar) -> Image:
    out = Image.empty(x.C, x.F, x.B, x.T, x.Tp, x.buf.device)
    call("idv_planar_to_image", x.ptr(), i(x.C), i(x.F), i(x.B * x.Tp), i(x.Jp), out.ptr(), ll(out.lo_off), stream_ptr())
    return out


def to_image_repeat(x: Planar, n: int) -> Image:
    """to_image(repeat_batch(x, n)) in one pass: each utterance n times in a row, straight into the split image."""
    out = Image.empty(x.C, x.F, x.B * n, x.T, x.Tp, x.buf.device)
    call("idv_planar_to_image_repeat", x.ptr(), i(x.C), i(x.F), i(x.B), i(x.Tp), i(x.Jp), i(n), out.ptr(), ll(out.lo_off),
         i(out.Jp), stream_ptr())
    return out


def to_planar(x: Image) -> Planar:
    out = Planar.empty(x.C, x.F, x.B, x.T, x.Tp, x.buf.device)
    call("idv_image_to_planar", x.ptr(), ll(x.lo_off), i(x.C), i(x.F), i(x.B * x.Tp), i(x.Jp), out.ptr(), stream_ptr())
    return out


class KImage:
    """K-major split image  kimg[hi|lo][nplanes/8][Jp][8]  (octet o = planes 8o..8o+7): the activation operand of the
    bf16x3 point-wise contractions (LSTM projection, DFT, dense)."""

    __slots__ = ("buf", "nplanes", "Jp", "lo_off")

    def __init__(self, nplanes: int, Jp: int, device):
        assert nplanes % 8 == 0
        plane = nplanes // 8 * Jp * 8
        self.nplanes, self.Jp, self.lo_off = nplanes, Jp, plane + IMG_SLACK
        self.buf = torch.empty(bucket(2 * IMG_SLACK + self.lo_off + plane), dtype=torch.int16, device=device)

    def ptr(self, octet: int = 0):
        return L._P(self.buf.data_ptr() + 2 * IMG_SLACK + 16 * octet * self.Jp)

    @property
    def lo_slots(self) -> int:
        return self.lo_off // 8

    @classmethod
    def from_planes(cls, x_ptr, nvalid: int, J: int, Jp: int, device, pad_to: int = 8):
        nplanes = (nvalid + pad_to - 1) // pad_to * pad_to
        out = cls(nplanes, Jp, device)
        call("idv_planar_to_kimage", x_ptr, i(nvalid), i(nplanes), i(J), i(Jp), out.ptr(), ll(out.lo_off), stream_ptr())
        return out


def pack_pw_bf16(w):
    M, K = w.shape
    L.lib().idv_pw_bf16_wfrag_bytes.restype = L._L
    wf = torch.empty(int(L.lib().idv_pw_bf16_wfrag_bytes(i(M), i(K))), dtype=torch.uint8, device=w.device)
    call("idv_pack_pw_bf16", p(w.contiguous()), i(M), i(K), p(wf), stream_ptr())
    return wf


def pw_bf16x3(kimg: KImage, octet0: int, K: int, wfrag16, bias, M: int, B: int, Tp: int, t_valid: int, out_ptr):
    call("idv_pw_bf16x3", kimg.ptr(octet0), ll(kimg.lo_slots), i(K), p(wfrag16), p(bias), out_ptr, i(M), i(B), i(Tp), i(kimg.Jp),
         i(t_valid), stream_ptr())


def pw_bf16x3_rows(kimg: KImage, K: int, wfrag16, bias, M: int, ldo: int, B: int, T: int, Tp: int, out_ptr):
    """out[(t*B + b)*ldo + m] = sum_k W[m][k] x[k][(b, t)] + bias[m] over the valid frames (row-major, as idv_lstm_bptt reads)."""
    call("idv_pw_bf16x3_rows", kimg.ptr(), ll(kimg.lo_slots), i(K), p(wfrag16), p(bias), out_ptr, i(M), i(ldo), i(B), i(T), i(Tp),
         i(kimg.Jp), stream_ptr())


def _dev_scratch(n: int, device, dtype=torch.float32, zero=False):
    return (torch.zeros if zero else torch.empty)(n, dtype=dtype, device=device)


# ----------------------------------------------------------------------------- packing
def mtiles_alloc(M: int) -> int:
    return (M + 127) // 128 * 4


def cbn_fold(moments, g_rr, g_ri, g_ii, b_r, b_i):
    """moments: [5, C] tensor (mean_r, mean_i, Vrr, Vri, Vii) -> fold [C, 6]."""
    C = g_rr.numel()
    fold = torch.empty(C, 6, dtype=torch.float32, device=g_rr.device)
    call("idv_cbn_fold", p(moments), p(g_rr), p(g_ri), p(g_ii), p(b_r), p(b_i), i(C), p(fold), stream_ptr())
    return fold


def pack_cconv(w_re, w_im, b_re, b_im, fold, cin_used: Optional[int] = None, transposed=False):
    """-> (wfrag, bias) for idv_cconv2d_fwd."""
    if transposed:
        cin_total, cout = w_re.shape[0], w_re.shape[1]
    else:
        cout, cin_total = w_re.shape[0], w_re.shape[1]
    cin_used = cin_total if cin_used is None else cin_used
    cck = L.lib().idv_cconv_cck(cin_used)
    ccp = (2 * cin_used + cck - 1) // cck * cck
    mt = mtiles_alloc(2 * cout)
    wfrag = torch.empty(mt * ccp * 5 * 64, dtype=torch.float32, device=w_re.device)
    bias = torch.empty(mt * 32, dtype=torch.float32, device=w_re.device)
    call("idv_pack_cconv", p(w_re.contiguous()), p(w_im.contiguous()), p(b_re.contiguous()), p(b_im.contiguous()),
         p(fold), i(cout), i(cin_total), i(cin_used), i(1 if transposed else 0), p(wfrag), p(bias), stream_ptr())
    return wfrag, bias


SKIP_ONCE = os.environ.get("IDV_SKIP_ONCE", "1") != "0"      # fp32 eval, repeated skips: skip half of a decoder conv once per utterance
GAUSS_FWD = os.environ.get("IDV_GAUSS_FWD", "1") != "0"      # A/B switches: forward / data gradient on cgemm_kernel instead
GAUSS_BWD = os.environ.get("IDV_GAUSS_BWD", "1") != "0"


def gauss_supported(c0: int, c1: int, cout: int, bwd: bool = False) -> bool:
    """fp32 mode: does the three-product (Gauss) contraction kernel serve this layer (csrc/cgemm_gauss.hip)?"""
    if not (GAUSS_BWD if bwd else GAUSS_FWD):
        return False
    return bool(L.lib().idv_cconv_gauss_supported(i(c0), i(c1), i(cout)))


def pack_cconv_gauss(w_re, w_im, b_re, b_im, fold, cin_used: Optional[int] = None, transposed=False, *, adjoint_of=None):
    """-> (wfrag3, epi, has_fold) for idv_cconv2d_gauss_fwd.  adjoint_of=(cout_adj, cin_total_adj, cin_used_adj, adj_transposed):
    the data-gradient operator (conjugate-transposed weights, no bias), arguments describing the adjoint as pack_cconv_adjoint."""
    if adjoint_of is not None:
        cout, cin_total, cin_used, transposed = adjoint_of
    else:
        if transposed:
            cin_total, cout = w_re.shape[0], w_re.shape[1]
        else:
            cout, cin_total = w_re.shape[0], w_re.shape[1]
        cin_used = cin_total if cin_used is None else cin_used
    lib = L.lib()
    lib.idv_cconv_gauss_wfrag_floats.restype = L._L
    wfrag = torch.empty(int(lib.idv_cconv_gauss_wfrag_floats(i(cout), i(cin_used))), dtype=torch.float32, device=w_re.device)
    epi = torch.empty(int(lib.idv_cconv_gauss_epi_rows(i(cout))) * 8, dtype=torch.float32, device=w_re.device)
    if adjoint_of is not None:
        # adjoint of a conv [Cout][Cin] is a transposed conv whose [Cin'][Cout'] weight IS that tensor (Cin' = Cout), and vice
        # versa: same memory, the other interpretation, W_i negated (idv_pack_cconv_adjoint does the same)
        call("idv_pack_cconv_gauss", p(w_re), p(w_im), p(None), p(None), p(None), i(cout), i(cin_total), i(cin_used),
             i(1 if transposed else 0), i(1), p(wfrag), p(epi), stream_ptr())
        return (wfrag, epi, 0) + _pack_wino_tw(w_re, w_im, cout, cin_total, cin_used, transposed, 1)
    w_re, w_im = w_re.contiguous(), w_im.contiguous()
    call("idv_pack_cconv_gauss", p(w_re), p(w_im), p(b_re.contiguous()), p(b_im.contiguous()), p(fold),
         i(cout), i(cin_total), i(cin_used), i(1 if transposed else 0), i(0), p(wfrag), p(epi), stream_ptr())
    return (wfrag, epi, (1 if fold is not None else 0)) + _pack_wino_tw(w_re, w_im, cout, cin_total, cin_used, transposed, 0)


# fp32 convs / transposed convs with Winograd-transformed frequency taps on top of the three-product form (csrc/cgemm_wino.hip):
# 7 instead of 10 real products per input channel and pair of rows.  IDV_WINO=0 (or ops.WINO = False) keeps cgemm_gauss.
WINO = os.environ.get("IDV_WINO", "1") != "0"
WINO_CFG = 4000000               # LAUNCH_LOG ids: WINO_CFG + (1000 if transposed) + idv_cconv_wino_config


def _pack_wino(w_re, w_im, cout: int, cin_total: int, cin_used: int, transposed: bool, conj: int):
    """Winograd-transformed Gauss planes of an operator (transposed = the OPERATOR's mode: a ComplexConvTranspose2d or the adjoint
    of a ComplexConv2d; a ComplexConv2d or the adjoint of a transposed one), flags exactly as idv_pack_cconv_gauss gets them."""
    if not WINO:
        return None
    n = int(_ll_fn("idv_cconv_wino_wfrag_floats")(i(1 if transposed else 0), i(cout), i(cin_used)))
    wf = torch.empty(n, dtype=torch.float32, device=w_re.device)
    call("idv_pack_cconv_wino", p(w_re), p(w_im), i(cout), i(cin_total), i(cin_used), i(1 if transposed else 0), i(conj), p(wf),
         stream_ptr())
    return wf


# ... and, for the transposed operators, with the TIME taps in Winograd form too (csrc/cgemm_tw.hip): 3 instead of 4 products per pair
# of output columns.  IDV_TW=0 (or ops.TW = False before the weights
# are packed) keeps cgemm_wino.  Measured at B = 64: dec0-3 38.9 -> 35.2 ms, headline 823 -> 864 utt/s on the same box.
TW = os.environ.get("IDV_TW", "1") != "0"
TW_CONV = os.environ.get("IDV_TW_CONV", "1") != "0"      # the conv form (csrc/cgemm_tw2.hip); 0: cgemm_wino's conv form
TW_CFG = 5000000                 # LAUNCH_LOG ids of a launch on the time-Winograd kernels: TW_CFG (+ 1: taps (x[t-1], x[t]); + 2: the conv form)


def _pack_wino_tw(w_re, w_im, cout: int, cin_total: int, cin_used: int, transposed: bool, conj: int):
    """-> (wino fragments | None, time-Winograd fragments | None): the tail of a gauss pack tuple (the time-Winograd fragments are
    those of the OPERATOR's form: csrc/cgemm_tw.hip for a transposed conv, csrc/cgemm_tw2.hip for a conv)."""
    wf = _pack_wino(w_re, w_im, cout, cin_total, cin_used, transposed, conj)
    if wf is None or not TW or (not transposed and not TW_CONV):
        return wf, None
    if transposed:
        tw = torch.empty(int(_ll_fn("idv_cconv_tw_wfrag_floats")(i(cout), i(cin_used))), dtype=torch.float32, device=w_re.device)
        call("idv_pack_cconv_tw", p(wf), i(cout), i(cin_used), p(tw), stream_ptr())
    else:
        tw = torch.empty(int(_ll_fn("idv_cconv_tw2_wfrag_floats")(i(cout), i(cin_used))), dtype=torch.float32, device=w_re.device)
        call("idv_pack_cconv_tw2", p(wf), i(cout), i(cin_used), p(tw), stream_ptr())
    return wf, tw


def _tw2_ok(gauss, x: Planar, c1: int, cout: int) -> bool:
    return (gauss is not None and TW and TW_CONV and WINO and c1 == 0 and len(gauss) > 4 and gauss[4] is not None and x.Jp % 4 == 0
            and bool(L.lib().idv_cconv_tw2_supported(i(x.C), i(cout), i(x.F))))


def _tw_ok(gauss, x: Planar, c1: int, cout: int, skip_jp: Optional[int]) -> bool:
    return (gauss is not None and TW and WINO and len(gauss) > 4 and gauss[4] is not None and x.Jp % 4 == 0
            and (skip_jp is None or skip_jp == x.Jp) and bool(L.lib().idv_cconv_tw_supported(i(x.C), i(c1), i(cout), i(x.F))))


def _wino_ok(gauss, transposed: bool, x: Planar, c1: int, cout: int, skip_jp: Optional[int]) -> bool:
    return (gauss is not None and WINO and len(gauss) > 3 and gauss[3] is not None and x.Jp % 4 == 0
            and (skip_jp is None or skip_jp == x.Jp)
            and bool(L.lib().idv_cconv_wino_supported(i(1 if transposed else 0), i(x.C), i(c1), i(cout), i(x.F))))


def pack_cconv_gauss_skip_part(w_re, w_im, c0: int):
    """Gauss operands of the SKIP half of a transposed conv [Cin][Cout][5][2] (input channels c0 .. Cin), no bias, no fold:
    the once-per-utterance addend of the repeated-skip decoder (see cconv2d(addend=...))."""
    wr, wi = w_re[c0:].contiguous(), w_im[c0:].contiguous()
    cin, cout = wr.shape[0], wr.shape[1]
    lib = L.lib()
    lib.idv_cconv_gauss_wfrag_floats.restype = L._L
    wfrag = torch.empty(int(lib.idv_cconv_gauss_wfrag_floats(i(cout), i(cin))), dtype=torch.float32, device=wr.device)
    epi = torch.empty(int(lib.idv_cconv_gauss_epi_rows(i(cout))) * 8, dtype=torch.float32, device=wr.device)
    call("idv_pack_cconv_gauss", p(wr), p(wi), p(None), p(None), p(None), i(cout), i(cin), i(cin), i(1), i(0), p(wfrag), p(epi),
         stream_ptr())
    return (wfrag, epi, 0) + _pack_wino_tw(wr, wi, cout, cin, cin, True, 0)


def bf16_supported(transposed: bool, c0: int, c1: int, skip_div: int, cout: int) -> bool:
    return bool(L.lib().idv_cconv_bf16_supported(i(1 if transposed else 0), i(c0), i(c1), i(skip_div), i(cout)))


def pack_cconv_bf16(w_re, w_im, fold, cin_used: Optional[int] = None, transposed=False):
    """-> split-bf16 weight fragments (uint8 tensor) for idv_cconv2d_bf16x3_fwd."""
    if transposed:
        cin_total, cout = w_re.shape[0], w_re.shape[1]
    else:
        cout, cin_total = w_re.shape[0], w_re.shape[1]
    cin_used = cin_total if cin_used is None else cin_used
    L.lib().idv_cconv_bf16_wfrag_bytes.restype = L._L
    nbytes = L.lib().idv_cconv_bf16_wfrag_bytes(i(cout), i(cin_used))
    wfrag = torch.empty(int(nbytes), dtype=torch.uint8, device=w_re.device)
    call("idv_pack_cconv_bf16", p(w_re.contiguous()), p(w_im.contiguous()), p(fold), i(cout), i(cin_total), i(cin_used),
         i(1 if transposed else 0), p(wfrag), stream_ptr())
    return wfrag


def pack_ctconv_c1(w_re, w_im, fold, cin_used: Optional[int] = None):
    """Split-bf16 fragments for the Cout = 1 transposed conv (idv_ctconv_c1_bf16x3_fwd)."""
    cin_total = w_re.shape[0]
    cin_used = cin_total if cin_used is None else cin_used
    L.lib().idv_ctconv_c1_wfrag_bytes.restype = L._L
    wfrag = torch.empty(int(L.lib().idv_ctconv_c1_wfrag_bytes(i(cin_used))), dtype=torch.uint8, device=w_re.device)
    call("idv_pack_ctconv_c1_bf16", p(w_re.contiguous()), p(w_im.contiguous()), p(fold), i(cin_total), i(cin_used), p(wfrag),
         stream_ptr())
    return wfrag


def ctconv_c1(x, wfrag_c1, bias, *, slope=None, skip=None) -> Planar:
    """Causal transposed conv with one output channel on the split-bf16 path; x / skip both Planar or both Image."""
    out = Planar.empty(1, 2 * x.F - 1, x.B, x.T, x.Tp, x.buf.device)
    c1 = skip.C if skip is not None else 0
    if LAUNCH_LOG is not None:
        macs = 4 * (x.C + c1) * 10 * x.B * x.T * x.F
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if isinstance(x, Image):
        if skip is not None and not isinstance(skip, Image):
            raise RuntimeError("ctconv_c1: x and skip must have the same format")
        call("idv_ctconv_c1_img_fwd", x.ptr(), ll(x.lo_slots), i(x.C), skip.ptr() if skip is not None else p(None),
             ll(skip.lo_slots if skip is not None else 0), i(c1), p(wfrag_c1), p(bias), p(slope), out.ptr(), i(x.F), i(x.B),
             i(x.Tp), i(x.Jp), i(x.T), stream_ptr())
    else:
        call("idv_ctconv_c1_bf16x3_fwd", x.ptr(), i(x.C), skip.ptr() if skip is not None else p(None), i(c1),
             i(skip.Jp if skip is not None else 0), p(wfrag_c1), p(bias), p(slope), out.ptr(), i(x.F), i(x.B), i(x.Tp), i(x.Jp),
             i(x.T), stream_ptr())
    if LAUNCH_LOG is not None:
        ev1.record()
        LAUNCH_LOG.append((-98 if isinstance(x, Image) else -99, macs, ev0, ev1))
    return out


def pack_pw(w, bias):
    M, K = w.shape
    mt = mtiles_alloc(M)
    ks = (K + 7) // 8 * 4
    wfrag = torch.empty(mt * ks * 64, dtype=torch.float32, device=w.device)
    bout = torch.empty(mt * 32, dtype=torch.float32, device=w.device)
    call("idv_pack_pw", p(w.contiguous()), p(None if bias is None else bias.contiguous()), i(M), i(K), p(wfrag),
         p(bout), stream_ptr())
    return wfrag, bout


# ----------------------------------------------------------------------------- forward ops
# When set to a list, every cconv2d launch appends (config id, algorithmic MACs, start event, end event);
# bench.py uses it to time the dominant kernel with events on the launching stream.
LAUNCH_LOG = None

# Eval-mode sub-batch pipelining (model/pvae_module.py DCCRN_.forward): number of HIP streams a batch is split over.
# Concurrent queues are only safe because the library is built without packed-fp32 VALU code (__graft_entry__.build,
# -fno-slp-vectorize): on MI355X v_pk_fma_f32 / v_pk_mul_f32 results of one kernel came out wrong in 16-lane slices
# while an MFMA-saturating kernel of another stream shared its SIMDs.
STREAM_SPLIT = int(os.environ.get("IDV_STREAM_SPLIT", "1"))     # opt-in: 2 = sub-batch pipelining
CONCURRENT = os.environ.get("IDV_CONCURRENT", "0") != "0"      # opt-in: ops.concurrent really uses several streams
STREAM_STAGGER = os.environ.get("IDV_STREAM_STAGGER", "1") != "0"   # part k+1 starts when part k reaches its LSTM
STREAM_STAGGER_BELOW = 1 << 30                   # parts below this size are staggered; measured +6 / +4 / +2 % at B = 64 / 96 / 128
STREAM_SPLIT_MIN_BATCH = 16                      # per-stream utterances below which launch overhead dominates
_SIDE_STREAMS = {}


def stream_split(batch: int) -> int:
    n = max(1, STREAM_SPLIT)
    while n > 1 and batch < n * STREAM_SPLIT_MIN_BATCH:
        n -= 1
    return n


def side_streams(n: int, device):
    key = (torch.device(device).index or 0)
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def repeat_batch(x: Planar, n: int) -> Planar:
    """Each utterance n times in a row (`skip.repeat_interleave(n, 0)`, reference pvae_module.py:2563-2567) as a new
    planar buffer, so the repeated-skip decoder can use the kernels that take x1_div == 1 sources (bf16x3 / images)."""
    out = Planar.empty(x.C, x.F, x.B * n, x.T, x.Tp, x.buf.device)
    dst = torch.as_strided(out.buf, (2, x.C, x.F, x.B, n, x.Tp),
                           (x.C * x.F * out.Jp, x.F * out.Jp, out.Jp, n * x.Tp, x.Tp, 1), SLACK)
    dst.copy_(x.planes().unsqueeze(4).expand(2, x.C, x.F, x.B, n, x.Tp))      # planes(): [2, C, F, B, Tp]
    return out


def concurrent(fns, device=None):
    """Run independent callables (e.g. the frozen clean / noise encoders and the noisy encoder of the NSVAE step) on
    separate HIP streams and join: their latency-bound phases (the per-step recurrences) overlap.  Results may be
    used on the current stream afterwards; they return to their stream's pool, whose next use through this function
    starts behind everything enqueued on the current stream until then."""
    if not CONCURRENT:
        return [fn() for fn in fns]
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    main = torch.cuda.current_stream(device)
    streams = side_streams(len(fns), device)
    ready = torch.cuda.Event()
    ready.record(main)
    outs = []
    for fn, st in zip(fns, streams):
        st.wait_event(ready)
        with torch.cuda.stream(st):
            outs.append(fn())
    for st in streams:
        main.wait_stream(st)
    hand_over(outs, main)
    return outs


def hand_over(obj, stream):
    """Tell the caching allocator that `stream` reads every tensor reachable from `obj` (nested tuples / lists, the planar
    buffer a reference-shaped view carries as `_idv`): blocks that came from another stream's pool are then not handed
    out again before `stream` has passed the point where they are released.  Explicit cross-stream ownership instead of an
    argument about what the other stream does next."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
        pl = getattr(obj, "_idv", None)
        if pl is not None and isinstance(getattr(pl, "buf", None), torch.Tensor) and pl.buf.is_cuda:
            pl.buf.record_stream(stream)
    elif isinstance(obj, (Planar, Image)):
        obj.buf.record_stream(stream)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            hand_over(o, stream)


class CoopTimeout(RuntimeError):
    """A cooperative (spin-synchronised) recurrence ran into its spin bound: its outputs are NaN-poisoned."""


def coop_check(sync: bool = True, clear: bool = True) -> None:
    """Report a cooperative recurrence's time-out AT THE OPERATION THAT OWNS IT: call after the forward / step whose result is
    about to be used (inference.py's entry points, bench.py's timed region and smoke() do).  `sync` synchronises the current
    stream first -- the status word is written by the device when the kernel aborts.  Raises CoopTimeout and, with `clear`,
    acknowledges the fault so that the next cooperative launch runs again; without it every cooperative entry keeps refusing
    with status -3 (sticky, like a device error), csrc/coop.hpp."""
    if sync:
        torch.cuda.current_stream().synchronize()
    if L.lib().idv_coop_last_status(1 if clear else 0) != 0:
        raise CoopTimeout("a cooperative LSTM recurrence timed out on this device (a sibling workgroup never became resident, e.g. "
                          "a CU-masked or shared GPU): results computed since the last check are NaN-poisoned"
                          + ("" if clear else "; cooperative launches are refused until ops.coop_clear()"))


def coop_clear() -> bool:
    """Acknowledge a cooperative time-out without raising; True if there was one."""
    return L.lib().idv_coop_last_status(1) != 0


# Train-mode moment sums: the conv epilogues spread their atomic adds over this many replicas of the [Cout][5] sums
# (idv_cconv2d_fwd / idv_cconv2d_gauss_fwd stats_work); IDV_STATS_REP=1 keeps one set
STATS_REP = int(os.environ.get("IDV_STATS_REP", "32"))


def _stats_work(stats, cout: int):
    if stats is None or STATS_REP < 2:
        return None
    return torch.zeros(STATS_REP, cout, 5, dtype=torch.float64, device=stats.device)


def cconv2d(x: Planar, wfrag, bias, cout: int, *, transposed=False, causal=True, slope=None, skip: Optional[Planar] = None,
            skip_div: int = 1, stats: Optional[torch.Tensor] = None, out: Optional[Planar] = None,
            wfrag_bf16: Optional[torch.Tensor] = None, image: str = "", adjoint_time: bool = False, gauss=None,
            addend: Optional[Planar] = None, addend_div: int = 1):
    """(causal_)ComplexConv2d / (causal_)ComplexConvTranspose2d forward on planar activations.
    image="also" / "only": the exact-fp32 kernel additionally / only writes a split-bf16 image -> (Planar|None, Image)."""
    Fout = 2 * x.F - 1 if transposed else (x.F - 1) // 2 + 1
    if causal:
        t_out = x.T
    else:
        t_out = x.T + 1 if transposed else x.T - 1
    tshift = -1 if (causal or transposed) else 0
    if adjoint_time:
        # transposed conv reading (x[t+1], x[t]): with conjugate-transposed weights the adjoint (data gradient) of
        # the causal conv; all T frames are produced (x[T] is the next utterance's zero guard column)
        assert transposed and causal
        tshift = 0
    if out is None and image != "only":
        out = Planar.empty(cout, Fout, x.B, t_out, x.Tp, x.buf.device)
    c1 = skip.C if skip is not None else 0
    if image:
        assert stats is None and skip_div == 1 and wfrag_bf16 is None
        img = Image.empty(cout, Fout, x.B, t_out, x.Tp, x.buf.device)
        call("idv_cconv2d_fwd_img", x.ptr(), i(x.C), skip.ptr() if skip is not None else p(None), i(c1),
             i(skip.Jp if skip is not None else 0), p(wfrag), p(bias), p(slope), out.ptr() if out is not None else p(None),
             img.ptr(), ll(img.lo_off), i(1 if transposed else 0), i(tshift), i(cout), i(x.F), i(x.B), i(x.Tp), i(x.Jp),
             i(t_out), stream_ptr())
        return out, img
    if LAUNCH_LOG is not None:
        cfg = L.lib().idv_cconv_config(i(1 if transposed else 0), i(x.C + c1), i(cout), i(x.F))
        # algorithmic MACs: 4 real convolutions of the reference, kernel 5x2, per kept output position
        # (transposed: per INPUT position, each input feeds 5x2 taps)
        pos = x.B * x.T * (x.F if transposed else Fout)
        macs = 4 * (x.C + c1) * cout * 10 * pos
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if wfrag_bf16 is not None:
        if LAUNCH_LOG is not None:
            cfg = -(1000000 + L.lib().idv_cconv_bf16_config(i(1 if transposed else 0), i(cout), i(x.F)))
        swork = _stats_work(stats, cout)
        call("idv_cconv2d_bf16x3_fwd", x.ptr(), i(x.C), skip.ptr() if skip is not None else p(None), i(c1),
             i(skip.Jp if skip is not None else 0), i(skip_div), p(wfrag_bf16), p(bias), p(slope), out.ptr(), p(stats), p(swork),
             i(STATS_REP), i(1 if transposed else 0), i(tshift), i(cout), i(x.F), i(x.B), i(x.Tp), i(x.Jp), i(t_out), stream_ptr())
        if LAUNCH_LOG is not None:
            ev1.record()
            LAUNCH_LOG.append((cfg, macs, ev0, ev1))
        return out
    if transposed and skip_div == 1 and _tw_ok(gauss, x, c1, cout, skip.Jp if skip is not None else None):
        # fp32: Winograd-transformed frequency and time taps (csrc/cgemm_tw.hip)
        if LAUNCH_LOG is not None:
            cfg = TW_CFG + (1 if tshift else 0)
        swork = _stats_work(stats, cout)
        call("idv_ctconv2d_tw_fwd", x.ptr(), i(x.C), skip.ptr() if skip is not None else p(None), i(c1), p(gauss[4]), p(gauss[1]),
             i(gauss[2]), p(slope), out.ptr(), p(stats), p(swork), i(STATS_REP), i(tshift), i(cout), i(x.F), i(x.B), i(x.Tp), i(x.Jp),
             i(t_out), addend.ptr() if addend is not None else p(None), i(addend_div), i(addend.Jp if addend is not None else 0),
             stream_ptr())
    elif not transposed and addend is None and _tw2_ok(gauss, x, c1, cout):
        # fp32 conv: Winograd-transformed frequency and time taps (csrc/cgemm_tw2.hip)
        if LAUNCH_LOG is not None:
            cfg = TW_CFG + 2 + (1 if tshift else 0)
        swork = _stats_work(stats, cout)
        call("idv_cconv2d_tw_fwd", x.ptr(), i(x.C), p(gauss[4]), p(gauss[1]), i(gauss[2]), p(slope), out.ptr(), p(stats), p(swork),
             i(STATS_REP), i(tshift), i(cout), i(x.F), i(x.B), i(x.Tp), i(x.Jp), i(t_out), stream_ptr())
    elif skip_div == 1 and _wino_ok(gauss, transposed, x, c1, cout, skip.Jp if skip is not None else None):
        # fp32: Winograd-transformed frequency taps on top of the three products (csrc/cgemm_wino.hip)
        if LAUNCH_LOG is not None:
            cfg = WINO_CFG + (1000 if transposed else 0) + L.lib().idv_cconv_wino_config(i(1 if transposed else 0), i(x.C + c1), i(cout))
        swork = _stats_work(stats, cout)
        call("idv_cconv2d_wino_fwd", x.ptr(), i(x.C), skip.ptr() if skip is not None else p(None), i(c1), p(gauss[3]), p(gauss[1]),
             i(gauss[2]), p(slope), out.ptr(), p(stats), p(swork), i(STATS_REP), i(1 if transposed else 0), i(tshift), i(cout), i(x.F),
             i(x.B), i(x.Tp), i(x.Jp), i(t_out), addend.ptr() if addend is not None else p(None), i(addend_div),
             i(addend.Jp if addend is not None else 0), stream_ptr())
    elif gauss is not None:
        # fp32: three real products per complex product (csrc/cgemm_gauss.hip); gauss = (wfrag3, epi, has_fold[, wino fragments])
        if LAUNCH_LOG is not None:
            cfg = L.lib().idv_cconv_gauss_config(i(1 if transposed else 0), i(x.C + c1), i(cout), i(x.F))
        swork = _stats_work(stats, cout)
        call("idv_cconv2d_gauss_fwd", x.ptr(), i(x.C), skip.ptr() if skip is not None else p(None), i(c1),
             i(skip.Jp if skip is not None else 0), i(skip_div), p(gauss[0]), p(gauss[1]), i(gauss[2]), p(slope), out.ptr(),
             p(stats), p(swork), i(STATS_REP), i(1 if transposed else 0), i(tshift), i(cout), i(x.F), i(x.B), i(x.Tp), i(x.Jp), i(t_out),
             addend.ptr() if addend is not None else p(None), i(addend_div), i(addend.Jp if addend is not None else 0), stream_ptr())
    else:
        swork = _stats_work(stats, cout)
        call("idv_cconv2d_fwd", x.ptr(), i(x.C), skip.ptr() if skip is not None else p(None), i(c1),
             i(skip.Jp if skip is not None else 0), i(skip_div), p(wfrag), p(bias), p(slope), out.ptr(), p(stats), p(swork),
             i(STATS_REP), i(1 if transposed else 0), i(tshift), i(cout), i(x.F), i(x.B), i(x.Tp), i(x.Jp), i(t_out), stream_ptr())
    if LAUNCH_LOG is not None:
        ev1.record()
        LAUNCH_LOG.append((cfg, macs, ev0, ev1))
    return out


def cconv2d_img(x, wfrag_bf16, bias, cout: int, *, transposed=False, causal=True, slope=None, skip=None,
                want_planar=False, want_image=True, adjoint=False):
    """Eval-mode conv / transposed conv on the split-bf16 path with image and/or planar sources (x and skip must share
    one format) -> (Planar or None, Image or None).  adjoint: the data-gradient form (time taps reversed, all T frames kept,
    see cconv_dgrad)."""
    src_img = isinstance(x, Image)
    if skip is not None and isinstance(skip, Image) != src_img:
        raise RuntimeError("cconv2d_img: x and skip must have the same format")
    Fout = 2 * x.F - 1 if transposed else (x.F - 1) // 2 + 1
    t_out = x.T if causal else (x.T + 1 if transposed else x.T - 1)
    tshift = -1 if (causal or transposed) else 0
    if adjoint:
        assert causal
        tshift, t_out = 0, x.T
    dev = x.buf.device
    outp = Planar.empty(cout, Fout, x.B, t_out, x.Tp, dev) if want_planar else None
    outi = Image.empty(cout, Fout, x.B, t_out, x.Tp, dev) if want_image else None
    c1 = skip.C if skip is not None else 0
    if LAUNCH_LOG is not None:
        cfg = -(100000000 + L.lib().idv_cconv_img_config(i(1 if src_img else 0), i(1 if transposed else 0), i(x.C + c1),
                                                          i(cout), i(x.F)))
        pos = x.B * x.T * (x.F if transposed else Fout)
        macs = 4 * (x.C + c1) * cout * 10 * pos
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    call("idv_cconv2d_img_fwd", i(1 if src_img else 0), x.ptr(), ll(x.lo_slots if src_img else 0), i(x.C),
         skip.ptr() if skip is not None else p(None), ll(skip.lo_slots if (skip is not None and src_img) else 0), i(c1),
         p(wfrag_bf16), p(bias), p(slope), outp.ptr() if outp is not None else p(None),
         outi.ptr() if outi is not None else p(None), ll(outi.lo_off if outi is not None else 0),
         i(1 if transposed else 0), i(tshift), i(cout), i(x.F), i(x.B), i(x.Tp), i(x.Jp), i(t_out), stream_ptr())
    if LAUNCH_LOG is not None:
        ev1.record()
        LAUNCH_LOG.append((cfg, macs, ev0, ev1))
    return outp, outi


def cconv2d_img_train(x: Image, wfrag_bf16, bias, cout: int, stats, *, transposed=False, skip: Optional[Image] = None) -> Planar:
    """Training forward of a causal conv block from split images: planar fp32 y + the batch-norm moments (`stats`)."""
    Fout = 2 * x.F - 1 if transposed else (x.F - 1) // 2 + 1
    out = Planar.empty(cout, Fout, x.B, x.T, x.Tp, x.buf.device)
    c1 = skip.C if skip is not None else 0
    if LAUNCH_LOG is not None:
        cfg = -(100000000 + L.lib().idv_cconv_img_config(i(1), i(1 if transposed else 0), i(x.C + c1), i(cout), i(x.F)))
        macs = 4 * (x.C + c1) * cout * 10 * x.B * x.T * (x.F if transposed else Fout)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    swork = _stats_work(stats, cout)
    call("idv_cconv2d_img_train_fwd", x.ptr(), ll(x.lo_slots), i(x.C), skip.ptr() if skip is not None else p(None),
         ll(skip.lo_slots if skip is not None else 0), i(c1), p(wfrag_bf16), p(bias), out.ptr(), p(stats), p(swork), i(STATS_REP),
         i(1 if transposed else 0), i(cout), i(x.F), i(x.B), i(x.Tp), i(x.Jp), i(x.T), stream_ptr())
    if LAUNCH_LOG is not None:
        ev1.record()
        LAUNCH_LOG.append((cfg, macs, ev0, ev1))
    return out


def pw_gemm(x_ptr, K: int, wfrag, bias, M: int, B: int, Tp: int, Jp: int, t_valid: int, out_ptr, *, slope=None,
            swap=False, ldo=0):
    call("idv_pw_gemm", x_ptr, i(K), p(wfrag), p(bias), p(slope), out_ptr, i(M), i(B), i(Tp), i(Jp), i(t_valid),
         i(1 if swap else 0), i(ldo), stream_ptr())


def pack_lstm(sd_get, H: int, K: int, layer: int, device):
    """sd_get(name) -> tensor for names like 'lstm_re.weight_ih_l0'.  Returns (wih, bih, whh)."""
    g = lambda n: sd_get(n).contiguous()
    l = layer
    M = 8 * H
    mt = mtiles_alloc(M)
    ks = (K + 7) // 8 * 4
    wih = torch.empty(mt * ks * 64, dtype=torch.float32, device=device)
    bih = torch.empty(mt * 32, dtype=torch.float32, device=device)
    call("idv_pack_lstm_ih", p(g(f"lstm_re.weight_ih_l{l}")), p(g(f"lstm_re.bias_ih_l{l}")), p(g(f"lstm_re.bias_hh_l{l}")),
         p(g(f"lstm_im.weight_ih_l{l}")), p(g(f"lstm_im.bias_ih_l{l}")), p(g(f"lstm_im.bias_hh_l{l}")), i(H), i(K),
         p(wih), p(bih), stream_ptr())
    whh = torch.empty(4 * 4 * H * H, dtype=torch.float32, device=device)
    call("idv_pack_lstm_hh", p(g(f"lstm_re.weight_hh_l{l}")), p(g(f"lstm_im.weight_hh_l{l}")), i(H), p(whh), stream_ptr())
    wih16 = None
    if L.lib().idv_lstm_proj_bf16_supported(i(H), i(K)) and (layer == 0 or (4 * H) % 256 == 0):
        L.lib().idv_lstm_ih_bf16_bytes.restype = L._L
        wih16 = torch.empty(int(L.lib().idv_lstm_ih_bf16_bytes(i(H), i(K))), dtype=torch.uint8, device=device)
        call("idv_pack_lstm_ih_bf16", p(g(f"lstm_re.weight_ih_l{l}")), p(g(f"lstm_im.weight_ih_l{l}")), i(H), i(K), p(wih16),
             stream_ptr())
    # layer 1 at H = 128: W_ih ([4H][H] like W_hh) in the recurrence's fragment order too, for the two-layer cooperative
    # launch of the fp32 evaluation (csrc/lstm_stack2_f32.hip)
    wih_hh = None
    if layer == 1 and K == H and H == 128:
        wih_hh = torch.empty(4 * 4 * H * H, dtype=torch.float32, device=device)
        call("idv_pack_lstm_hh", p(g(f"lstm_re.weight_ih_l{l}")), p(g(f"lstm_im.weight_ih_l{l}")), i(H), p(wih_hh), stream_ptr())
    return wih, bih, whh, wih16, wih_hh


# bf16x3 mode, H = 384 / 768: one persistent cooperative launch per layer (csrc/lstm_pers.hip) instead of one launch per
# time step; IDV_LSTM_PERSISTENT=0 keeps the per-step kernels
LSTM_PERSISTENT = os.environ.get("IDV_LSTM_PERSISTENT", "1") != "0"
# fp32 evaluation, H = 128: both layers in one cooperative launch (csrc/lstm_stack2_f32.hip); 0 keeps one launch per layer
LSTM_STACK2 = os.environ.get("IDV_LSTM_STACK2", "1") != "0"


def clstm(x: Planar, packed0, packed1, H: int) -> Planar:
    """ComplexLSTM forward: x planar with C*F = K feature planes per part -> planar [2][H][Jp] (F = 1)."""
    K = x.C * x.F
    out = Planar.empty(H, 1, x.B, x.T, x.Tp, x.buf.device)
    nwork = L.lib().idv_clstm_work_floats(i(H), i(x.B), i(x.T), i(x.Jp))
    work = torch.empty(bucket(int(nwork)), dtype=torch.float32, device=x.buf.device)
    flags = 1 if PRECISION == "bf16x3" else 0
    if not LSTM_PERSISTENT:
        flags |= 8
    if (flags & 1) and packed0[3] is not None:
        # layer-0 input projection on the bf16 MFMA: K-major split image of the 2K input planes, then G into `work`
        kimg = KImage.from_planes(x.ptr(), 2 * K, x.B * x.Tp, x.Jp, x.buf.device)
        call("idv_lstm_proj_bf16x3", kimg.ptr(), ll(kimg.lo_slots), i(K), p(packed0[3]), p(packed0[1]), p(work), i(H), i(x.B),
             i(x.T), i(x.Tp), i(x.Jp), stream_ptr())
        flags |= 2
    call("idv_clstm_fwd2", x.ptr(), i(K), p(packed0[0]), p(packed0[1]), p(packed0[2]), p(packed1[0]), p(packed1[1]),
         p(packed1[2]), p(packed1[4] if (LSTM_STACK2 and len(packed1) > 4) else None), i(H), i(x.B), i(x.T), i(x.Tp), i(x.Jp), p(work), out.ptr(),
         i(flags), p(packed1[3] if (flags & 1) else None), stream_ptr())
    return out


def cdense(x: Planar, packed_r, packed_i, M: int, C_out: int, F_out: int) -> Planar:
    """ComplexDense on a planar [2][K][Jp] activation -> planar [2][C_out][F_out][Jp] (M = C_out*F_out)."""
    K = x.C * x.F
    out = Planar.empty(C_out, F_out, x.B, x.T, x.Tp, x.buf.device)
    if PRECISION == "bf16x3" and K % 64 == 0 and len(packed_r) > 2 and packed_r[2] is not None:
        kimg = KImage.from_planes(x.ptr(), 2 * K, x.B * x.Tp, x.Jp, x.buf.device)
        for ri, pk in enumerate((packed_r, packed_i)):
            pw_bf16x3(kimg, ri * K // 8, K, pk[2], pk[1], M, x.B, x.Tp, x.T, out.ptr(ri * C_out))
        return out
    for ri, pk in enumerate((packed_r, packed_i)):
        pw_gemm(x.ptr(ri * x.C), K, pk[0], pk[1], M, x.B, x.Tp, x.Jp, x.T, out.ptr(ri * C_out))
    return out


class DftPlan:
    """Windowed DFT / inverse DFT matrices for one (n_fft, win, hop, T), packed for idv_pw_gemm."""

    def __init__(self, n_fft, win, hop, T, device):
        self.n_fft, self.win, self.hop, self.T = n_fft, win, hop, T
        self.F = n_fft // 2 + 1
        F = self.F
        w_fwd = torch.empty(2 * F, win, dtype=torch.float32, device=device)
        w_inv = torch.empty(win, 2 * F, dtype=torch.float32, device=device)
        self.env_inv = torch.empty(n_fft + hop * (T - 1), dtype=torch.float32, device=device)
        call("idv_make_dft", i(n_fft), i(win), i(hop), i(T), p(w_fwd), p(w_inv), p(self.env_inv), stream_ptr())
        self.fwd = pack_pw(w_fwd, None)
        self.inv = pack_pw(w_inv, None)
        self.fwd16 = pack_pw_bf16(w_fwd)          # split-bf16 fragments of the same matrices (bf16x3 mode)
        self.inv16 = pack_pw_bf16(w_inv)
        self._w_fwd, self._w_inv = w_fwd, w_inv
        self._fwd_T = self._inv_T = None

    def fwd_T(self):
        """Packed transposed DFT matrix [win][2F]: the adjoint (signal gradient) of the STFT contraction."""
        if self._fwd_T is None:
            self._fwd_T = pack_pw(self._w_fwd.t().contiguous(), None)
        return self._fwd_T

    def inv_T(self):
        """Packed transposed inverse-DFT matrix [2F][win]: the adjoint (spectrum gradient) of the ISTFT contraction."""
        if self._inv_T is None:
            self._inv_T = pack_pw(self._w_inv.t().contiguous(), None)
        return self._inv_T


def stft(x: torch.Tensor, plan: DftPlan, Tp: Optional[int] = None) -> Planar:
    """STFT.forward: x [B, L] -> planar [2][1][F][Jp]."""
    B, Lx = x.shape
    T = 1 + Lx // plan.hop
    assert T == plan.T, "DftPlan built for another length"
    Tp = Tp or T + 1
    x = x.contiguous()
    if PRECISION == "bf16x3":
        Jp = Planar.jp_for(B, Tp)
        kimg = KImage((plan.win + 63) // 64 * 64, Jp, x.device)
        call("idv_stft_frames_kimage", p(x), i(B), i(Lx), i(plan.n_fft), i(plan.win), i(plan.hop), i(T), kimg.ptr(),
             ll(kimg.lo_off), i(Tp), i(Jp), stream_ptr())
        out = Planar.empty(1, plan.F, B, T, Tp, x.device)
        pw_bf16x3(kimg, 0, plan.win, plan.fwd16, None, 2 * plan.F, B, Tp, T, out.ptr())
        return out
    fr = Planar.empty(1, plan.win // 2, B, T, Tp, x.device)      # [win][Jp] scratch (2*C*F = win planes)
    assert plan.win % 2 == 0
    call("idv_stft_frames", p(x), i(B), i(Lx), i(plan.n_fft), i(plan.win), i(plan.hop), i(T), fr.ptr(), i(Tp), i(fr.Jp),
         stream_ptr())
    out = Planar.empty(1, plan.F, B, T, Tp, x.device)
    pw_gemm(fr.ptr(), plan.win, plan.fwd[0], plan.fwd[1], 2 * plan.F, B, Tp, fr.Jp, T, out.ptr())
    return out


def istft(spec: Planar, plan: DftPlan) -> torch.Tensor:
    """ISTFT.forward: planar [2][1][F][Jp] -> y [B, hop*(T-1)]."""
    B, T = spec.B, spec.T
    fr = Planar.empty(1, plan.win // 2, B, T, spec.Tp, spec.buf.device)
    if PRECISION == "bf16x3":
        kimg = KImage.from_planes(spec.ptr(), 2 * plan.F, B * spec.Tp, spec.Jp, spec.buf.device, pad_to=64)
        pw_bf16x3(kimg, 0, 2 * plan.F, plan.inv16, None, plan.win, B, spec.Tp, T, fr.ptr())
    else:
        pw_gemm(spec.ptr(), 2 * plan.F, plan.inv[0], plan.inv[1], plan.win, B, spec.Tp, spec.Jp, T, fr.ptr())
    y = torch.empty(B, plan.hop * (T - 1), dtype=torch.float32, device=spec.buf.device)
    call("idv_istft_ola", fr.ptr(), p(plan.env_inv), i(B), i(plan.n_fft), i(plan.win), i(plan.hop), i(T), i(spec.Tp),
         i(fr.Jp), p(y), stream_ptr())
    return y


def mask_apply(mask: Planar, X: Planar, x_div: int = 1):
    """-> (pred planar, pred complex64 [B, F, T])."""
    B, F, T = mask.B, mask.F, mask.T
    pred = Planar.empty(1, F, B, T, mask.Tp, mask.buf.device)
    pc = torch.empty(B, F, T, 2, dtype=torch.float32, device=mask.buf.device)
    call("idv_mask_apply", mask.ptr(), X.ptr(), i(x_div), i(X.Jp), pred.ptr(), p(pc), i(F), i(B), i(T), i(mask.Tp),
         i(mask.Jp), stream_ptr())
    return pred, torch.view_as_complex(pc)


def msd(a: Planar, ca0: int, b: Planar, cb0: int, C: int) -> torch.Tensor:
    """One term of residual_loss (nsvae_loss.py:363-446): mean over [B, C, F, T, 2] of (a[:, ca0:ca0+C] - b[:, cb0:cb0+C])^2."""
    work = torch.empty(3, dtype=torch.float64, device=a.buf.device)
    out = torch.empty(1, dtype=torch.float32, device=a.buf.device)
    call("idv_msd", a.ptr(), i(a.C), i(ca0), i(a.Jp), b.ptr(), i(b.C), i(cb0), i(b.Jp), i(C), i(a.F), i(a.B), i(a.Tp), i(a.T),
         p(work), p(out), stream_ptr())
    return out[0]


def planar_to_complex(act: Planar) -> torch.Tensor:
    pc = torch.empty(act.B, act.F, act.T, 2, dtype=torch.float32, device=act.buf.device)
    call("idv_planar_to_complex", act.ptr(), p(pc), i(act.F), i(act.B), i(act.T), i(act.Tp), i(act.Jp), stream_ptr())
    return torch.view_as_complex(pc)


def cbn_train(act: Planar, stats: torch.Tensor, bn, slope, first_call: bool, momentum: float = 0.9):
    """Finish a train-mode conv block: stats (filled by cconv2d(stats=...)) -> batch moments, running
    buffers (in place, on the module's own buffers), then y = PReLU(Z x + s) in place on `act`."""
    C = act.C
    dev = act.buf.device
    moments = torch.empty(5, C, dtype=torch.float32, device=dev)
    fold = torch.empty(C, 6, dtype=torch.float32, device=dev)
    count = float(act.B) * act.F * act.T
    if BN_SYNC is not None:
        count = count * BN_SYNC(stats)
    call("idv_cbn_finalize", p(stats), d(count), p(bn.gamma_rr), p(bn.gamma_ri), p(bn.gamma_ii), p(bn.beta_r), p(bn.beta_i),
         i(C), i(1 if first_call else 0), f(momentum), p(bn.running_mean_real), p(bn.running_mean_imag), p(bn.Vrr), p(bn.Vri),
         p(bn.Vii), p(moments), p(fold), stream_ptr())
    call("idv_cbn_apply_prelu", act.ptr(), p(fold), p(slope), i(C), i(act.F), i(act.B), i(act.Tp), i(act.Jp), i(act.T),
         stream_ptr())
    return moments


def reparam(lat: Planar, off, zdim: int, eps_r, eps_i, ns: int) -> Planar:
    """reparameterization on a planar latent; off = (miu, log_sigma, delta) channel offsets."""
    B, T = lat.B, lat.T
    z = Planar.empty(zdim, 1, B * ns, T, lat.Tp, lat.buf.device)
    call("idv_reparam", lat.ptr(), i(lat.C), i(off[0]), i(off[1]), i(off[2]), i(zdim), p(eps_r.contiguous()),
         p(eps_i.contiguous()), i(ns), i(B), i(T), i(lat.Tp), i(lat.Jp), z.ptr(), i(z.Jp), stream_ptr())
    return z


# ----------------------------------------------------------------------------- losses
def check_dev_f32(t: torch.Tensor, name: str, device=None):
    """The kernels dereference raw pointers: a CPU tensor (e.g. straight out of a DataLoader) must fail here, not on the GPU."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise L.IdvError(f"{name}: expected a CUDA (ROCm) tensor, got {type(t).__name__} on "
                         f"{getattr(t, 'device', '?')}; the HIP hot path has no CPU fallback")
    if device is not None and t.device != torch.device(device):
        raise L.IdvError(f"{name}: on {t.device}, expected {device}")


def sisnr(source: torch.Tensor, est: torch.Tensor, src_div: int = 1) -> torch.Tensor:
    B, Ln = est.shape
    check_dev_f32(source, "source", est.device)
    if source.dtype != torch.float32 or est.dtype != torch.float32 or source.shape[-1] < Ln:
        raise L.IdvError("sisnr: float32 tensors with source length >= estimate length expected")
    assert source.stride(-1) == 1 and est.stride(-1) == 1
    work = torch.empty(3 * B, dtype=torch.float64, device=est.device)
    out = torch.empty(1, dtype=torch.float32, device=est.device)
    call("idv_sisnr", p(source), i(source.stride(0)), i(src_div), p(est), i(est.stride(0)), i(B), i(Ln), p(work), p(out),
         stream_ptr())
    return out[0]


def recon_loss(pred_c: torch.Tensor, ori: torch.Tensor, ori_div: int = 1):
    """pred_c: complex64 [B,F,T] (contiguous); ori: [B/ori_div, F, T, 2] real, any strides -> (cpx, mag)."""
    pr = torch.view_as_real(pred_c)
    assert pr.is_contiguous()
    check_dev_f32(ori, "ori_cpx_stft", pr.device)
    if ori.dtype != torch.float32:
        raise L.IdvError("recon_loss: float32 STFT expected")
    B, F, T, _ = pr.shape
    if tuple(ori.shape[1:]) != (F, T, 2):
        raise L.IdvError(f"recon_loss: STFT shape {tuple(ori.shape)} does not match the prediction {tuple(pr.shape)}")
    work = torch.empty(3, dtype=torch.float64, device=pr.device)
    out = torch.empty(2, dtype=torch.float32, device=pr.device)
    sb, sf, st, sr = ori.stride()
    call("idv_recon_loss", p(pr), p(ori), ll(sb), ll(sf), ll(st), ll(sr), i(ori_div), i(B), i(F), i(T), p(work), p(out),
         stream_ptr())
    return out[0], out[1]


def ckl(q1: Planar, off1, q2: Optional[Planar], off2, zdim: int, eps: float) -> torch.Tensor:
    work = torch.empty(3, dtype=torch.float64, device=q1.buf.device)
    out = torch.empty(1, dtype=torch.float32, device=q1.buf.device)
    o2 = off2 if q2 is not None else (0, 0, 0)
    call("idv_ckl", q1.ptr(), i(q1.C), i(q1.Jp), i(off1[0]), i(off1[1]), i(off1[2]),
         q2.ptr() if q2 is not None else p(None), i(q2.C if q2 is not None else 0), i(q2.Jp if q2 is not None else 0),
         i(o2[0]), i(o2[1]), i(o2[2]), i(zdim), f(eps), i(q1.B), i(q1.T), i(q1.Tp), p(work), p(out), stream_ptr())
    return out[0]


def mi_estimate(lat: Planar, off, z: Planar, zdim: int, ns: int, eps: float):
    """Minibatch mutual information (pretrain_pvaes_loss.py:129-159) -> (scalar, work); `work` holds d MI / d log q for mi_bwd."""
    lib = L.lib()
    lib.idv_mi_work_floats.restype = L._L
    if (z.C, z.B, z.T, z.Tp) != (zdim, lat.B * ns, lat.T, lat.Tp):
        raise ValueError(f"mutual_information: samples {(z.C, z.B, z.T)} do not belong to a posterior {(zdim, lat.B, lat.T)} x {ns}")
    work = torch.empty(int(lib.idv_mi_work_floats(i(lat.B), i(ns), i(lat.T), i(zdim))), dtype=torch.float32, device=lat.buf.device)
    acc = torch.empty(1, dtype=torch.float64, device=lat.buf.device)
    out = torch.empty(1, dtype=torch.float32, device=lat.buf.device)
    call("idv_mi_fwd", lat.ptr(), i(lat.C), i(lat.Jp), i(off[0]), i(off[1]), i(off[2]), z.ptr(), i(z.Jp), i(zdim), i(ns), i(lat.B),
         i(lat.T), i(lat.Tp), f(eps), p(work), p(acc), p(out), stream_ptr())
    return out[0], work


def mi_bwd(lat: Planar, off, z: Planar, zdim: int, ns: int, eps: float, work, gout, dlat: Optional[Planar], dz: Optional[Planar]):
    call("idv_mi_bwd", lat.ptr(), i(lat.C), i(lat.Jp), i(off[0]), i(off[1]), i(off[2]), z.ptr(), i(z.Jp), i(zdim), i(ns), i(lat.B),
         i(lat.T), i(lat.Tp), f(eps), p(work), p(gout), dlat.ptr() if dlat is not None else p(None),
         dz.ptr() if dz is not None else p(None), stream_ptr())


def miu_dist(q1: Planar, off1: int, q2: Planar, off2: int, zdim: int) -> torch.Tensor:
    work = torch.empty(3 * 2 * zdim, dtype=torch.float64, device=q1.buf.device)
    out = torch.empty(1, dtype=torch.float32, device=q1.buf.device)
    call("idv_miu_dist", q1.ptr(), i(q1.C), i(q1.Jp), i(off1), q2.ptr(), i(q2.C), i(q2.Jp), i(off2), i(zdim), i(q1.B), i(q1.T),
         i(q1.Tp), p(work), p(out), stream_ptr())
    return out[0]


def cbn_stats(act: Planar) -> torch.Tensor:
    """Moment sums [C][5] (double) of a planar activation, for a stand-alone train-mode batch norm."""
    stats = torch.zeros(act.C, 5, dtype=torch.float64, device=act.buf.device)
    call("idv_cbn_stats", act.ptr(), i(act.C), i(act.F), i(act.B), i(act.Tp), i(act.Jp), i(act.T), p(stats), stream_ptr())
    return stats


def cbn_apply(act: Planar, fold: torch.Tensor, slope=None):
    call("idv_cbn_apply_prelu", act.ptr(), p(fold), p(slope), i(act.C), i(act.F), i(act.B), i(act.Tp), i(act.Jp), i(act.T),
         stream_ptr())
    return act


# ----------------------------------------------------------------------------- backward (gradient) operators
# Thin wrappers over the idv_*_bwd entries; autograd.py strings them into torch.autograd.Function classes.
_LL = L._L


def _ll_fn(name):
    fn = getattr(L.lib(), name)
    fn.restype = _LL
    return fn


def like(x: Planar, C: Optional[int] = None, F: Optional[int] = None, zero=False) -> Planar:
    return Planar.empty(x.C if C is None else C, x.F if F is None else F, x.B, x.T, x.Tp, x.buf.device, zero=zero)


def rewrap(buf: torch.Tensor, x: Planar, C: Optional[int] = None, F: Optional[int] = None) -> Planar:
    """A Planar with x's geometry over another flat buffer (e.g. an incoming gradient)."""
    return Planar(buf, x.C if C is None else C, x.F if F is None else F, x.B, x.T, x.Tp, x.Jp)


def pack_cconv_adjoint(w_re, w_im, cout: int, cin_total: int, cin_used: int, transposed: bool):
    """Fragments of the adjoint operator (data gradient), see idv_pack_cconv_adjoint; arguments describe the adjoint."""
    cck = L.lib().idv_cconv_cck(cin_used)
    ccp = (2 * cin_used + cck - 1) // cck * cck
    mt = mtiles_alloc(2 * cout)
    wfrag = torch.empty(mt * ccp * 5 * 64, dtype=torch.float32, device=w_re.device)
    bias = torch.empty(mt * 32, dtype=torch.float32, device=w_re.device)
    call("idv_pack_cconv_adjoint", p(w_re), p(w_im), i(cout), i(cin_total), i(cin_used), i(1 if transposed else 0), p(wfrag),
         p(bias), stream_ptr())
    return wfrag, bias


def pack_cconv_bf16_adjoint(w_re, w_im, cout: int, cin_total: int, cin_used: int, transposed: bool):
    """Split-bf16 fragments of the adjoint operator (bf16x3 training), see idv_pack_cconv_bf16_adjoint."""
    L.lib().idv_cconv_bf16_wfrag_bytes.restype = L._L
    wfrag = torch.empty(int(L.lib().idv_cconv_bf16_wfrag_bytes(i(cout), i(cin_used))), dtype=torch.uint8, device=w_re.device)
    call("idv_pack_cconv_bf16_adjoint", p(w_re), p(w_im), i(cout), i(cin_total), i(cin_used), i(1 if transposed else 0),
         p(wfrag), stream_ptr())
    return wfrag


_ZBIAS = {}


def zero_bias(cout: int, device):
    key = (mtiles_alloc(2 * cout), str(device))
    if key not in _ZBIAS:
        _ZBIAS[key] = torch.zeros(key[0] * 32, dtype=torch.float32, device=device)
        torch.cuda.current_stream(device).synchronize()       # filled once; later users may sit on other streams
    return _ZBIAS[key]


def cconv_dgrad(dy: Planar, wfrag, bias, cout_adj: int, fwd_transposed: bool, causal: bool, wfrag_bf16=None, gauss=None) -> Planar:
    """Data gradient of a causal_complex_conv2d / causal_ComplexConvTranspose2d: the adjoint operator on idv_cconv2d_fwd
    (transposed conv reading (dy[t+1], dy[t]) / conv reading (dy[t], dy[t+1]); all T frames kept: column T+1 is the next
    utterance's zero guard column).  wfrag_bf16: run it on the split-bf16 kernel instead (bf16x3 training mode)."""
    adj_transposed = not fwd_transposed
    Fout = 2 * dy.F - 1 if adj_transposed else (dy.F - 1) // 2 + 1
    # causal blocks: time taps reversed (tshift 0), T frames in, T frames out.  Non-causal blocks (padding (2, 0)): the adjoint of
    # the conv (T -> T - 1 frames) is the transposed conv's own tap order (dx[t] = W0' dy[t] + W1' dy[t-1], tshift -1) on T frames;
    # the adjoint of the transposed conv (T -> T + 1) is the non-causal conv's (dx[t] = W0' dy[t] + W1' dy[t+1], tshift 0) on T frames
    if causal:
        tshift_adj, t_out = 0, dy.T
    elif adj_transposed:
        tshift_adj, t_out = -1, dy.T + 1
    else:
        tshift_adj, t_out = 0, dy.T - 1
    if t_out + 1 > dy.Tp:
        raise RuntimeError("cconv_dgrad: the input of the block had more frames than the buffer's columns per utterance")
    out = Planar.empty(cout_adj, Fout, dy.B, t_out, dy.Tp, dy.buf.device)
    if LAUNCH_LOG is not None:
        cfg = L.lib().idv_cconv_config(i(1 if adj_transposed else 0), i(dy.C), i(cout_adj), i(dy.F))
        macs = 4 * dy.C * cout_adj * 10 * dy.B * dy.T * (dy.F if adj_transposed else Fout)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if wfrag_bf16 is not None:
            cfg = -(1000000 + L.lib().idv_cconv_bf16_config(i(1 if adj_transposed else 0), i(cout_adj), i(dy.F)))
        ev0.record()
    if wfrag_bf16 is None and adj_transposed and _tw_ok(gauss, dy, 0, cout_adj, None):
        if LAUNCH_LOG is not None:
            cfg = TW_CFG + (1 if tshift_adj else 0)
        call("idv_ctconv2d_tw_fwd", dy.ptr(), i(dy.C), p(None), i(0), p(gauss[4]), p(gauss[1]), i(0), p(None), out.ptr(), p(None), p(None),
             i(0), i(tshift_adj), i(cout_adj), i(dy.F), i(dy.B), i(dy.Tp), i(dy.Jp), i(t_out), p(None), i(1), i(0), stream_ptr())
    elif wfrag_bf16 is None and not adj_transposed and _tw2_ok(gauss, dy, 0, cout_adj):
        if LAUNCH_LOG is not None:
            cfg = TW_CFG + 2 + (1 if tshift_adj else 0)
        call("idv_cconv2d_tw_fwd", dy.ptr(), i(dy.C), p(gauss[4]), p(gauss[1]), i(0), p(None), out.ptr(), p(None), p(None), i(0),
             i(tshift_adj), i(cout_adj), i(dy.F), i(dy.B), i(dy.Tp), i(dy.Jp), i(t_out), stream_ptr())
    elif wfrag_bf16 is None and _wino_ok(gauss, adj_transposed, dy, 0, cout_adj, None):
        # the data gradient on the Winograd kernels (csrc/cgemm_wino.hip): adjoint of a conv = a transposed conv and vice versa
        if LAUNCH_LOG is not None:
            cfg = WINO_CFG + (1000 if adj_transposed else 0) + L.lib().idv_cconv_wino_config(i(1 if adj_transposed else 0), i(dy.C), i(cout_adj))
        call("idv_cconv2d_wino_fwd", dy.ptr(), i(dy.C), p(None), i(0), p(gauss[3]), p(gauss[1]), i(0), p(None), out.ptr(), p(None), p(None),
             i(0), i(1 if adj_transposed else 0), i(tshift_adj), i(cout_adj), i(dy.F), i(dy.B), i(dy.Tp), i(dy.Jp), i(t_out), p(None), i(1),
             i(0), stream_ptr())
    elif gauss is not None and wfrag_bf16 is None:
        if LAUNCH_LOG is not None:
            cfg = L.lib().idv_cconv_gauss_config(i(1 if adj_transposed else 0), i(dy.C), i(cout_adj), i(dy.F))
        call("idv_cconv2d_gauss_fwd", dy.ptr(), i(dy.C), p(None), i(0), i(0), i(1), p(gauss[0]), p(gauss[1]), i(0), p(None),
             out.ptr(), p(None), p(None), i(0), i(1 if adj_transposed else 0), i(tshift_adj), i(cout_adj), i(dy.F), i(dy.B), i(dy.Tp),
             i(dy.Jp), i(t_out), p(None), i(1), i(0), stream_ptr())
    elif wfrag_bf16 is not None:
        call("idv_cconv2d_bf16x3_fwd", dy.ptr(), i(dy.C), p(None), i(0), i(0), i(1), p(wfrag_bf16), p(bias), p(None), out.ptr(), p(None),
             p(None), i(0), i(1 if adj_transposed else 0), i(tshift_adj), i(cout_adj), i(dy.F), i(dy.B), i(dy.Tp), i(dy.Jp), i(t_out),
             stream_ptr())
    else:
        call("idv_cconv2d_fwd", dy.ptr(), i(dy.C), p(None), i(0), i(0), i(1), p(wfrag), p(bias), p(None), out.ptr(), p(None), p(None),
             i(0), i(1 if adj_transposed else 0), i(tshift_adj), i(cout_adj), i(dy.F), i(dy.B), i(dy.Tp), i(dy.Jp), i(t_out), stream_ptr())
    if LAUNCH_LOG is not None:
        ev1.record()
        LAUNCH_LOG.append((cfg, macs, ev0, ev1))
    return out


_WORK = {}
WGRAD_BF16_CFG = -96     # ... of wgrad_bf16_kernel + its unpack (bf16x3 training mode)
WGRAD_BF16 = os.environ.get("IDV_WGRAD_BF16", "1") != "0"
WGRAD_GAUSS_CFG = -95    # ... of the three-product fp32 weight gradient (wgrad_combine + wgrad_kernel x 3 products + unpack)
WGRAD_CFG = -97          # LAUNCH_LOG id of the conv weight-gradient kernel (wgrad_kernel<5, 2, 1, 1, 4, 1, 16, 2> + its unpack)


def _scratch(n: int, device, tag="w") -> torch.Tensor:
    """Grow-only fp32 scratch per (device, stream, tag): weight-gradient partials, reused across layers and steps."""
    key = (str(device), torch.cuda.current_stream().cuda_stream, tag)
    t = _WORK.get(key)
    if t is None or t.numel() < n:
        t = torch.empty(bucket(n), dtype=torch.float32, device=device)
        _WORK[key] = t
    return t


def cconv_wgrad(x: Planar, ci_off: int, dy: Planar, cout: int, cin_total: int, transposed: bool, causal: bool, dw_re, dw_im):
    tshift = -1 if (causal or transposed) else 0
    cs, cl = (x.C, cout) if transposed else (cout, x.C)
    # bf16x3 training mode: the wide layers (>= 32 planes on both sides) on the split-bf16 MFMA kernel; the one-channel
    # ends (a handful of planes against a 128 x 32 plane tile) stay on the exact-fp32 kernel
    bf16 = PRECISION == "bf16x3" and WGRAD_BF16 and 2 * cs >= 32 and 2 * cl >= 32
    # exact fp32 with three real contractions per complex channel pair (Gauss, csrc/wgrad.hip) where both sides are wide enough
    gauss = not bf16 and bool(L.lib().idv_cconv_wgrad_gauss_supported(i(cs), i(cl)))
    if gauss:
        n = int(_ll_fn("idv_cconv_wgrad_gauss_work_floats")(i(x.C), i(cout), i(1 if transposed else 0), i(x.F), i(x.B), i(x.Tp),
                                                            i(x.Jp), i(dy.Jp)))
    else:
        n = int(_ll_fn("idv_cconv_wgrad_bf16_work_floats" if bf16 else "idv_cconv_wgrad_work_floats")(i(cs), i(cl), i(x.B), i(x.Tp)))
    work = _scratch(n, x.buf.device)
    if LAUNCH_LOG is not None:
        macs = 4 * x.C * cout * 10 * x.B * x.T * (x.F if transposed else dy.F)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    call("idv_cconv2d_bwd_weight_bf16x3" if bf16 else ("idv_cconv2d_bwd_weight_gauss" if gauss else "idv_cconv2d_bwd_weight"),
         x.ptr(), i(x.C), i(ci_off), dy.ptr(), i(cout), i(cin_total), i(1 if transposed else 0),
         i(tshift), i(x.F), i(x.B), i(x.Tp), i(x.Jp), i(dy.Jp), p(work), ll(work.numel()), p(dw_re), p(dw_im), stream_ptr())
    if LAUNCH_LOG is not None:
        ev1.record()
        LAUNCH_LOG.append((WGRAD_BF16_CFG if bf16 else (WGRAD_GAUSS_CFG if gauss else WGRAD_CFG), macs, ev0, ev1))


def cconv_bias_grad(dy: Planar):
    stats = cbn_stats(dy)
    db_re = torch.empty(dy.C, dtype=torch.float32, device=dy.buf.device)
    db_im = torch.empty_like(db_re)
    call("idv_cconv2d_bwd_bias", p(stats), i(dy.C), p(db_re), p(db_im), stream_ptr())
    return db_re, db_im


def train_image_ok(C: int) -> bool:
    """bf16x3 training: is an activation with C channels worth a split image (the image conv kernels take it as a source)?"""
    return PRECISION == "bf16x3" and IMAGE_TRAIN and C % 8 == 0 and 2 * C >= 64


def cbn_apply_to(y: Planar, fold, slope, want_image: bool = False):
    """-> z (Planar), or (z, Image of z) with want_image (one pass writes both)."""
    out = like(y)
    if want_image:
        img = Image.empty(y.C, y.F, y.B, y.T, y.Tp, y.buf.device)
        call("idv_cbn_apply_prelu_to_img", y.ptr(), p(fold), p(slope), i(y.C), i(y.F), i(y.B), i(y.Tp), i(y.Jp), i(y.T), out.ptr(),
             img.ptr(), ll(img.lo_off), stream_ptr())
        return out, img
    call("idv_cbn_apply_prelu_to", y.ptr(), p(fold), p(slope), i(y.C), i(y.F), i(y.B), i(y.Tp), i(y.Jp), i(y.T), out.ptr(),
         stream_ptr())
    return out


# Data-parallel training hook (parallel.py): when set, called on the per-channel moment sums of every train-mode
# ComplexBatchNormal (forward: [C][5] doubles, backward: [C][8] doubles) to all-reduce them over the ranks; returns
# the world size so the element count becomes the global one.
BN_SYNC = None


def cbn_bwd(dz: Planar, y: Planar, fold, moments, bn, slope, count: float, want_image: bool = False):
    """-> (dy, d gamma_rr, d gamma_ri, d gamma_ii, d beta_r, d beta_i, dslope[1]); want_image: dy is (Planar, Image)."""
    C, dev = y.C, y.buf.device
    sums = torch.empty(C, 8, dtype=torch.float64, device=dev)
    call("idv_cbn_bwd_reduce", dz.ptr(), y.ptr(), p(fold), p(slope), i(C), i(y.F), i(y.B), i(y.Tp), i(y.Jp), i(y.T), p(sums),
         stream_ptr())
    world = 1
    if BN_SYNC is not None:
        world = BN_SYNC(sums)
        count = count * world
    coef = torch.empty(C, 12, dtype=torch.float32, device=dev)
    g = [torch.empty(C, dtype=torch.float32, device=dev) for _ in range(5)]
    dslope = torch.zeros(1, dtype=torch.float32, device=dev)
    call("idv_cbn_bwd_finalize", p(sums), d(count), p(moments), p(bn[0]), p(bn[1]), p(bn[2]), i(C), p(coef), p(g[0]), p(g[1]),
         p(g[2]), p(g[3]), p(g[4]), p(dslope if slope is not None else None), f(1.0 / world), stream_ptr())
    dy = like(y)
    if want_image:
        img = Image.empty(y.C, y.F, y.B, y.T, y.Tp, dev)
        call("idv_cbn_bwd_apply_img", dz.ptr(), y.ptr(), p(fold), p(coef), p(slope), i(C), i(y.F), i(y.B), i(y.Tp), i(y.Jp), i(y.T),
             dy.ptr(), img.ptr(), ll(img.lo_off), stream_ptr())
        return ((dy, img), *g, dslope)
    call("idv_cbn_bwd_apply", dz.ptr(), y.ptr(), p(fold), p(coef), p(slope), i(C), i(y.F), i(y.B), i(y.Tp), i(y.Jp), i(y.T),
         dy.ptr(), stream_ptr())
    return (dy, *g, dslope)


def cbn_finalize(stats, count: float, bn_mod, first_call: bool, momentum: float, update_running: bool = True):
    """stats -> (moments [5, C], fold [C, 6]); running buffers of bn_mod updated in place."""
    C = bn_mod.C
    dev = stats.device
    if BN_SYNC is not None:
        count = count * BN_SYNC(stats)
    moments = torch.empty(5, C, dtype=torch.float32, device=dev)
    fold = torch.empty(C, 6, dtype=torch.float32, device=dev)
    rb = (bn_mod.running_mean_real, bn_mod.running_mean_imag, bn_mod.Vrr, bn_mod.Vri, bn_mod.Vii) if update_running else (None,) * 5
    call("idv_cbn_finalize", p(stats), d(count), p(bn_mod.gamma_rr), p(bn_mod.gamma_ri), p(bn_mod.gamma_ii), p(bn_mod.beta_r),
         p(bn_mod.beta_i), i(C), i(1 if first_call else 0), f(momentum), p(rb[0]), p(rb[1]), p(rb[2]), p(rb[3]), p(rb[4]),
         p(moments), p(fold), stream_ptr())
    return moments, fold


def pw_wgrad(dout_ptr, M: int, Jp_d: int, x_ptr, K: int, Jp_x: int, J: int, dw: torch.Tensor, *, shift=0, rowmap=0, H=0,
             accumulate=False):
    n = int(_ll_fn("idv_pw_wgrad_work_floats")(i(M), i(K), i(J)))
    work = _scratch(n, dw.device)
    # bf16x3 training mode: the split-bf16 kernel for the big contractions (LSTM projections); tiny ones stay fp32
    bf16 = PRECISION == "bf16x3" and WGRAD_BF16 and M >= 64 and K >= 64
    call("idv_pw_bwd_weight_bf16x3" if bf16 else "idv_pw_bwd_weight", dout_ptr, i(M), i(Jp_d), x_ptr, i(K), i(Jp_x), i(J), i(shift), p(work), ll(work.numel()), p(dw),
         i(dw.stride(0)), i(rowmap), i(H), i(1 if accumulate else 0), stream_ptr())


def planar_rowsum(x_ptr, M: int, Jp: int, J: int, out: torch.Tensor, accumulate=False):
    call("idv_planar_rowsum", x_ptr, i(M), i(Jp), i(J), i(1 if accumulate else 0), p(out), stream_ptr())


def mask_apply_bwd(mask: Planar, X: Planar, x_div: int, dpred: Optional[Planar], dpred_c, want_dx: bool = False):
    dm = like(mask)
    dX = like(X) if want_dx else None
    call("idv_mask_apply_bwd", mask.ptr(), X.ptr(), i(x_div), i(X.Jp), dpred.ptr() if dpred is not None else p(None),
         p(dpred_c), i(mask.F), i(mask.B), i(mask.T), i(mask.Tp), i(mask.Jp), dm.ptr(), dX.ptr() if dX is not None else p(None),
         stream_ptr())
    return dm, dX


def complex_to_planar(xc: torch.Tensor, Tp: Optional[int] = None) -> Planar:
    """interleaved float [B, F, T, 2] -> planar [2][1][F][Jp]"""
    B, F, T, _ = xc.shape
    out = Planar.empty(1, F, B, T, Tp or T + 1, xc.device)
    call("idv_complex_to_planar", p(xc.contiguous()), out.ptr(), i(F), i(B), i(T), i(out.Tp), i(out.Jp), stream_ptr())
    return out


def istft_bwd(dy: torch.Tensor, spec: Planar, plan: "DftPlan") -> Planar:
    """d loss / d spec of ops.istft."""
    B, T = spec.B, spec.T
    dfr = Planar.empty(1, plan.win // 2, B, T, spec.Tp, dy.device)
    call("idv_istft_ola_bwd", p(dy.contiguous()), p(plan.env_inv), i(B), i(plan.n_fft), i(plan.win), i(plan.hop), i(T),
         i(spec.Tp), i(dfr.Jp), dfr.ptr(), stream_ptr())
    out = like(spec)
    wT = plan.inv_T()
    pw_gemm(dfr.ptr(), plan.win, wT[0], wT[1], 2 * plan.F, B, spec.Tp, dfr.Jp, T, out.ptr())
    return out


def stft_bwd(dX: Planar, plan: "DftPlan", Lx: int) -> torch.Tensor:
    """d loss / d signal of ops.stft."""
    B, T = dX.B, dX.T
    dfr = Planar.empty(1, plan.win // 2, B, T, dX.Tp, dX.buf.device)
    wT = plan.fwd_T()
    pw_gemm(dX.ptr(), 2 * plan.F, wT[0], wT[1], plan.win, B, dX.Tp, dX.Jp, T, dfr.ptr())
    dx = torch.empty(B, Lx, dtype=torch.float32, device=dX.buf.device)
    call("idv_stft_frames_bwd", dfr.ptr(), i(B), i(Lx), i(plan.n_fft), i(plan.win), i(plan.hop), i(T), i(dX.Tp), i(dfr.Jp),
         p(dx), stream_ptr())
    return dx


def reparam_bwd(lat: Planar, off, zdim: int, eps_r, eps_i, ns: int, dz: Planar, dlat: Planar):
    call("idv_reparam_bwd", lat.ptr(), i(lat.C), i(off[0]), i(off[1]), i(off[2]), i(zdim), p(eps_r), p(eps_i), i(ns), i(lat.B),
         i(lat.T), i(lat.Tp), i(lat.Jp), dz.ptr(), i(dz.Jp), dlat.ptr(), stream_ptr())
