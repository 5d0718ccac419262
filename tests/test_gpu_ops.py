"""Operator-level parity on the GPU: every idv_* kernel through the C ABI vs the CPU oracle
(same seeded inputs) and vs the golden vectors captured from the reference.
Tolerance: 2e-5 relative L2 per operator (fp32 MFMA, different summation order);
north_star asks for 1e-3 on the enhanced waveform."""
import numpy as np
import pytest
import torch

from conftest import relerr
from oracle import idccrn_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5
NFFT, HOP, WIN = 512, 100, 400


@pytest.fixture(scope="module")
def ops(amd):
    return amd.ops


def T_(a):
    return torch.from_numpy(np.asarray(a))


def _conv_case(ops, causal, transposed, cin, cout, F, T, B, seed, Tp_extra=1, fold=False, slope=None, skip_c=0,
               skip_div=1, gauss=False):
    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    cin_tot = cin + skip_c
    x = torch.randn(B, cin, F, T, 2, generator=g)
    shape = (cin_tot, cout, 5, 2) if transposed else (cout, cin_tot, 5, 2)
    wr, wi = torch.randn(shape, generator=g) * 0.2, torch.randn(shape, generator=g) * 0.2
    br, bi = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    xin = x
    sk = None
    if skip_c:
        sk = torch.randn(B // skip_div, skip_c, F, T, 2, generator=g)
        xin = torch.cat([x, sk.repeat_interleave(skip_div, dim=0)], dim=1)
    if transposed:
        want = O.complex_conv_transpose2d(xin, wr, br, wi, bi, (2, 1), (2, 0), causal)
    else:
        want = O.complex_conv2d(xin, wr, br, wi, bi, (2, 1), (2, 1) if causal else (2, 0), causal)
    fold_t = None
    if fold:
        C = cout
        mom = torch.stack([torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1,
                           0.5 + torch.rand(C, generator=g), 0.1 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)])
        gam = [1 + 0.1 * torch.randn(C, generator=g), torch.randn(C, generator=g), 1 + 0.1 * torch.randn(C, generator=g)]
        bet = [0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)]
        want = O.cbn_whiten_affine(want, mom[0], mom[1], mom[2], mom[3], mom[4], gam[0], gam[1], gam[2], bet[0], bet[1])
        fold_t = ops.cbn_fold(mom.to(dev), *[t.to(dev) for t in gam], *[t.to(dev) for t in bet])
    slope_t = None
    if slope is not None:
        slope_t = torch.tensor([slope], device=dev)
        want = O.prelu(want, torch.tensor(slope))
    Tp = max(T, want.shape[3]) + Tp_extra
    xp = ops.Planar.from_tensor5(x.to(dev), Tp)
    skp = ops.Planar.from_tensor5(sk.to(dev), Tp) if sk is not None else None
    if gauss:
        # the three-product kernel (csrc/cgemm_gauss.hip): BN fold applied in its epilogue
        assert ops.gauss_supported(cin, skip_c, cout)
        g3 = ops.pack_cconv_gauss(wr.to(dev), wi.to(dev), br.to(dev), bi.to(dev), fold_t, transposed=transposed)
        y = ops.cconv2d(xp, None, None, cout, transposed=transposed, causal=causal, slope=slope_t, skip=skp, skip_div=skip_div,
                        gauss=g3)
    else:
        wfrag, bias = ops.pack_cconv(wr.to(dev), wi.to(dev), br.to(dev), bi.to(dev), fold_t, transposed=transposed)
        y = ops.cconv2d(xp, wfrag, bias, cout, transposed=transposed, causal=causal, slope=slope_t, skip=skp, skip_div=skip_div)
    torch.cuda.synchronize()
    got = y.tensor5().cpu()
    assert got.shape == want.shape
    assert relerr(got, want) < TOL
    pl = y.planes()
    assert float(pl[..., 0].abs().max()) == 0.0                      # guard column stays zero
    if pl.shape[-1] > y.T + 1:
        assert float(pl[..., y.T + 1:].abs().max()) == 0.0
    return got


@pytest.mark.parametrize("causal,transposed,cin,cout,F,T,B", [
    (True, False, 4, 8, 17, 9, 2), (True, False, 1, 32, 33, 40, 3), (False, False, 4, 8, 17, 9, 2),
    (True, True, 6, 4, 9, 9, 2), (False, True, 6, 4, 9, 9, 2), (True, True, 8, 1, 17, 50, 3),
    (True, False, 32, 64, 65, 70, 2), (True, True, 64, 32, 9, 70, 2), (True, False, 2, 16, 5, 700, 1),
    (True, True, 16, 16, 5, 130, 5),
])
def test_cconv_plain(ops, causal, transposed, cin, cout, F, T, B):
    _conv_case(ops, causal, transposed, cin, cout, F, T, B, seed=1)


@pytest.mark.parametrize("causal,transposed,cin,cout,F,T,B,skip_c,skip_div,fold,slope", [
    (True, False, 32, 64, 129, 70, 2, 0, 1, False, None),      # conv, 5 output rows per tile, two co tiles per workgroup
    (True, False, 8, 40, 65, 33, 2, 0, 1, True, 0.2),          # conv, 3 rows per tile, ragged second co tile, fold + PReLU
    (True, False, 6, 16, 33, 40, 3, 0, 1, False, None),        # conv, 1 row x 4 column tiles, one co tile
    (True, False, 3, 8, 9, 21, 2, 0, 1, False, 0.1),           # odd channel count: ragged last K chunk
    (False, False, 4, 8, 17, 9, 2, 0, 1, False, None),         # non-causal taps (x[t], x[t+1])
    (True, False, 2, 16, 5, 700, 1, 0, 1, False, None),        # many column tiles
    (True, True, 64, 64, 9, 70, 2, 0, 1, False, None),         # transposed conv, two co tiles
    (True, True, 6, 4, 9, 9, 2, 0, 1, True, 0.3),
    (False, True, 6, 4, 9, 9, 2, 0, 1, False, None),
    (True, True, 16, 16, 5, 130, 5, 0, 1, False, None),
    (True, True, 8, 4, 9, 30, 2, 8, 1, False, None),           # skip concat
    (True, True, 8, 4, 9, 30, 6, 8, 3, True, 0.25),            # repeated skips (num_samples = 3): scalar staging path
    (True, True, 8, 36, 17, 30, 4, 5, 2, False, None),         # odd skip channel count + repeated skips
    (True, True, 32, 32, 33, 645, 2, 32, 1, False, 0.25),      # utterance-length columns, Tp = 646
])
def test_cconv_gauss(ops, causal, transposed, cin, cout, F, T, B, skip_c, skip_div, fold, slope):
    """The fp32 three-product (Gauss) contraction against the oracle's four real convolutions, every tile shape."""
    _conv_case(ops, causal, transposed, cin, cout, F, T, B, seed=21, fold=fold, slope=slope, skip_c=skip_c, skip_div=skip_div,
               gauss=True)


@pytest.mark.parametrize("transposed,causal,cin,cout,F,T,B,skip_c,fold,slope", [
    (True, True, 8, 40, 9, 37, 3, 0, True, 0.2),          # two co tiles x two column groups (ragged second co tile), odd row count
    (True, True, 16, 128, 5, 30, 2, 0, False, None),      # four co tiles x one column group
    (True, True, 8, 8, 6, 30, 2, 8, False, None),         # one co tile: NOT served (cgemm_gauss runs), skip concat
    (True, True, 8, 40, 6, 30, 2, 8, False, None),        # even row count, skip concat (second source)
    (True, True, 6, 36, 2, 9, 2, 0, False, 0.1),          # two input rows (one tile), channel count below the pack granularity
    (True, True, 7, 36, 3, 45, 1, 0, True, None),         # odd channel count: ragged last K chunk
    (True, False, 6, 40, 9, 9, 2, 0, False, None),        # non-causal taps (T + 1 output frames)
    (True, True, 32, 64, 33, 645, 2, 32, False, 0.25),    # utterance-length columns, Tp = 646, column tail
    (True, True, 256, 64, 17, 70, 2, 0, True, 0.25),      # a real layer width (dec3's channels)
    (False, True, 32, 64, 129, 70, 2, 0, False, None),    # conv: two co tiles x two column groups, odd output row count (65)
    (False, True, 8, 40, 65, 33, 2, 0, True, 0.2),        # conv: ragged second co tile, fold + PReLU
    (False, True, 72, 128, 17, 40, 3, 0, False, None),    # conv: four co tiles x one column group, 72 input channels (served from 64)
    (False, True, 3, 36, 9, 21, 2, 0, False, 0.1),        # conv: odd channel count, 5 output rows (three tiles, the last half empty)
    (False, False, 4, 40, 17, 9, 2, 0, False, None),      # conv: non-causal taps (x[t], x[t+1])
    (False, True, 2, 48, 5, 700, 1, 0, False, None),      # conv: many column tiles, 3 output rows
    (False, True, 128, 128, 33, 70, 2, 0, True, 0.25),    # conv: a real layer width (enc3), 17 output rows (last tile half empty)
    (False, True, 130, 160, 9, 21, 2, 0, False, 0.1),     # conv: ragged last K chunk, five co tiles (second workgroup: one valid tile)
    (False, False, 128, 128, 17, 9, 2, 0, False, None),   # conv: non-causal taps at a served width
    (False, True, 136, 128, 4, 30, 3, 0, True, None),     # conv: even input row count (Fout = 2: one tile)
    (False, True, 6, 36, 4, 30, 2, 0, False, None),       # conv below the served widths: cgemm_gauss runs
])
def test_cconv_wino(ops, transposed, causal, cin, cout, F, T, B, skip_c, fold, slope):
    """The conv / transposed conv with Winograd-transformed frequency taps (csrc/cgemm_wino.hip: F(2,3) on the even, F(2,2) on the
    odd taps, on top of the three complex products) against the oracle's four real convolutions AND against cgemm_gauss_kernel, in
    its workgroup shapes."""
    keep, keep_log = ops.WINO, ops.LAUNCH_LOG
    served = bool(ops.L.lib().idv_cconv_wino_supported(int(transposed), cin, skip_c, cout, F))
    try:
        ops.WINO = True
        ops.LAUNCH_LOG = []
        got = _conv_case(ops, causal, transposed, cin, cout, F, T, B, seed=61, fold=fold, slope=slope, skip_c=skip_c, gauss=True)
        assert bool([c for c, *_ in ops.LAUNCH_LOG if c >= ops.WINO_CFG]) == served, "Winograd kernel launched / not launched"
        ops.WINO = False
        ops.LAUNCH_LOG = []
        ref = _conv_case(ops, causal, transposed, cin, cout, F, T, B, seed=61, fold=fold, slope=slope, skip_c=skip_c, gauss=True)
        assert not [c for c, *_ in ops.LAUNCH_LOG if c >= ops.WINO_CFG]
    finally:
        ops.WINO, ops.LAUNCH_LOG = keep, keep_log
    e = relerr(got, ref)
    print(f"wino vs gauss kernel: {e:.2e}")
    assert e < 5e-6


@pytest.mark.parametrize("causal,cin,cout,F,T,B,skip_c,fold,slope", [
    (True, 8, 40, 9, 37, 3, 0, True, 0.2),          # two co tiles (ragged second), odd row count: a half tile in the even-row phase
    (True, 16, 128, 5, 30, 2, 0, False, None),      # four co tiles
    (True, 8, 40, 6, 30, 2, 8, False, None),        # even row count, skip concat (second source)
    (True, 6, 36, 2, 9, 2, 0, False, 0.1),          # two input rows (one tile), channel count below the pack granularity
    (True, 7, 36, 3, 45, 1, 0, True, None),         # odd channel count: ragged last K chunk; odd columns per utterance (Tp = 46)
    (False, 6, 40, 9, 9, 2, 0, False, None),        # non-causal taps (window column on the right)
    (True, 32, 64, 33, 645, 2, 32, False, 0.25),    # utterance-length columns, Tp = 646, column tail
    (True, 256, 64, 17, 70, 2, 0, True, 0.25),      # a real layer width (dec3's channels)
    (True, 16, 36, 4, 31, 3, 4, False, None),       # J = 96: not a multiple of 64; second source of 4 channels (ragged last chunk)
    (True, 16, 32, 9, 37, 2, 16, True, 0.25),       # ONE co tile (dec4's width: cgemm_wino does not serve it, cgemm_gauss is the reference)
    (True, 8, 40, 6, 30, 3, 0, False, 0.2),         # ODD number of columns (B = 3, Tp = 31: J = 93): the last column pair is half empty
])
def test_ctconv_time_winograd(ops, causal, cin, cout, F, T, B, skip_c, fold, slope):
    """The transposed conv with Winograd-transformed frequency AND time taps (csrc/cgemm_tw.hip: F(2,2) over pairs of output
    columns on top of cgemm_wino) against the oracle's four real convolutions and against cgemm_wino."""
    keep, keep_tw, keep_log = ops.WINO, ops.TW, ops.LAUNCH_LOG
    assert ops.L.lib().idv_cconv_tw_supported(cin, skip_c, cout, F)
    try:
        ops.WINO = ops.TW = True
        ops.LAUNCH_LOG = []
        got = _conv_case(ops, causal, True, cin, cout, F, T, B, seed=61, fold=fold, slope=slope, skip_c=skip_c, gauss=True)
        assert [c for c, *_ in ops.LAUNCH_LOG if c in (ops.TW_CFG, ops.TW_CFG + 1)], "time-Winograd kernel not launched"
        ops.TW = False
        ops.LAUNCH_LOG = []
        ref = _conv_case(ops, causal, True, cin, cout, F, T, B, seed=61, fold=fold, slope=slope, skip_c=skip_c, gauss=True)
        assert not [c for c, *_ in ops.LAUNCH_LOG if c in (ops.TW_CFG, ops.TW_CFG + 1)]
    finally:
        ops.WINO, ops.TW, ops.LAUNCH_LOG = keep, keep_tw, keep_log
    e = relerr(got, ref)
    print(f"time-Winograd vs cgemm_wino: {e:.2e}")
    assert e < 5e-6


@pytest.mark.parametrize("causal,cin,cout,F,T,B,fold,slope", [
    (True, 64, 64, 129, 70, 2, False, None),        # two co tiles, odd output row count (65): a half tile
    (True, 32, 64, 129, 70, 2, False, None),        # (32 input channels: served only with IDV_TW2_MIN_CIN=8)
    (True, 8, 40, 65, 33, 2, True, 0.2),            # ragged second co tile, fold + PReLU
    (True, 72, 128, 17, 40, 3, False, None),        # four co tiles, 72 input channels
    (True, 67, 36, 9, 21, 2, False, 0.1),           # odd channel count (ragged last K chunk), 5 output rows
    (False, 64, 40, 17, 9, 2, False, None),         # non-causal taps (x[t], x[t+1])
    (True, 64, 48, 5, 700, 1, False, None),         # many column tiles, 3 output rows, odd Tp
    (True, 128, 128, 33, 70, 2, True, 0.25),        # a real layer width (enc3)
    (True, 80, 32, 4, 30, 3, True, None),           # even input row count (Fout = 2: one tile), one co tile; odd column count (J = 93)
])
def test_cconv_time_winograd(ops, causal, cin, cout, F, T, B, fold, slope):
    """The conv with Winograd-transformed frequency AND time taps (csrc/cgemm_tw2.hip) against the oracle's four real convolutions and
    against the kernel it replaces (cgemm_wino's conv form or cgemm_gauss)."""
    keep = ops.WINO, ops.TW, ops.TW_CONV, ops.LAUNCH_LOG
    served = bool(ops.L.lib().idv_cconv_tw2_supported(cin, cout, F))     # (below 64 input channels: not by default; IDV_TW2_MIN_CIN=8 does)
    try:
        ops.WINO = ops.TW = ops.TW_CONV = True
        ops.LAUNCH_LOG = []
        got = _conv_case(ops, causal, False, cin, cout, F, T, B, seed=67, fold=fold, slope=slope, gauss=True)
        assert bool([c for c, *_ in ops.LAUNCH_LOG if c in (ops.TW_CFG + 2, ops.TW_CFG + 3)]) == served, "time-Winograd conv kernel launched / not launched"
        ops.TW_CONV = False
        ops.LAUNCH_LOG = []
        ref = _conv_case(ops, causal, False, cin, cout, F, T, B, seed=67, fold=fold, slope=slope, gauss=True)
        assert not [c for c, *_ in ops.LAUNCH_LOG if c in (ops.TW_CFG + 2, ops.TW_CFG + 3)]
    finally:
        ops.WINO, ops.TW, ops.TW_CONV, ops.LAUNCH_LOG = keep
    e = relerr(got, ref)
    print(f"time-Winograd conv vs the kernel it replaces: {e:.2e}")
    assert e < 5e-6


def test_ctconv_time_winograd_adjoint(ops):
    """The data gradient of a conv (= a transposed conv with conjugate-transposed weights, reversed time taps) on the
    time-Winograd kernels against cgemm_wino."""
    g = torch.Generator().manual_seed(12)
    dev = "cuda"
    keep, keep_tw, keep_log = ops.WINO, ops.TW, ops.LAUNCH_LOG
    try:
        for cin, cout, F in ((128, 136, 17), (40, 72, 9), (24, 72, 9)):
            wr, wi = (torch.randn(cout, cin, 5, 2, generator=g) * 0.1).to(dev), (torch.randn(cout, cin, 5, 2, generator=g) * 0.1).to(dev)
            Fo = (F - 1) // 2 + 1
            served = bool(ops.L.lib().idv_cconv_tw_supported(cout, 0, cin, Fo))      # (24 output channels: one co tile, not served)
            assert served == (cin > 32)
            dy = ops.Planar.from_tensor5(torch.randn(3, cout, Fo, 37, 2, generator=g).to(dev), 38)
            d = {}
            for tw in (False, True):
                ops.WINO, ops.TW = True, tw
                ga = ops.pack_cconv_gauss(wr, wi, None, None, None, adjoint_of=(cin, cout, cout, True))
                ops.LAUNCH_LOG = []
                d[tw] = ops.cconv_dgrad(dy, None, None, cin, False, True, gauss=ga).tensor5().cpu()
                assert bool([c for c, *_ in ops.LAUNCH_LOG if c in (ops.TW_CFG, ops.TW_CFG + 1)]) == (tw and served)
            assert relerr(d[True], d[False]) < 5e-6
    finally:
        ops.WINO, ops.TW, ops.LAUNCH_LOG = keep, keep_tw, keep_log


def test_cconv_gauss_stats_and_adjoint(ops):
    """Train-mode moment sums of the three-product kernel = those of cgemm_kernel, and its adjoint (data gradient) form =
    the adjoint on cgemm_kernel, for a conv and a transposed conv."""
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    for transposed, cin, cout, F in ((False, 8, 40, 33), (True, 40, 8, 9)):
        x = torch.randn(3, cin, F, 37, 2, generator=g)
        shape = (cin, cout, 5, 2) if transposed else (cout, cin, 5, 2)
        wr, wi = (torch.randn(shape, generator=g) * 0.2).to(dev), (torch.randn(shape, generator=g) * 0.2).to(dev)
        br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
        xp = ops.Planar.from_tensor5(x.to(dev), 38)
        wfrag, bias = ops.pack_cconv(wr, wi, br, bi, None, transposed=transposed)
        g3 = ops.pack_cconv_gauss(wr, wi, br, bi, None, transposed=transposed)
        st0 = torch.zeros(cout, 5, dtype=torch.float64, device=dev)
        st1 = torch.zeros_like(st0)
        y0 = ops.cconv2d(xp, wfrag, bias, cout, transposed=transposed, stats=st0)
        y1 = ops.cconv2d(xp, None, None, cout, transposed=transposed, stats=st1, gauss=g3)
        assert relerr(y1.tensor5().cpu(), y0.tensor5().cpu()) < 1e-5
        assert relerr(st1.cpu(), st0.cpu()) < 1e-5
        # adjoint: dy has the OUTPUT geometry; conjugate-transposed weights, reversed time taps
        dy = ops.Planar.from_tensor5(torch.randn(3, cout, y0.F, 37, 2, generator=g).to(dev), 38)
        wf, bz = ops.pack_cconv_adjoint(wr, wi, cin, cout, cout, not transposed)
        ga = ops.pack_cconv_gauss(wr, wi, None, None, None, adjoint_of=(cin, cout, cout, not transposed))
        d0 = ops.cconv_dgrad(dy, wf, bz, cin, transposed, True)
        d1 = ops.cconv_dgrad(dy, None, None, cin, transposed, True, gauss=ga)
        assert relerr(d1.tensor5().cpu(), d0.tensor5().cpu()) < 1e-5


def test_cconv_wino_stats_and_adjoint(ops):
    """Training forms of the Winograd kernels at served widths: train-mode moment sums (STATS variants) and the adjoint /
    data-gradient operators (adjoint of a conv = transposed conv and vice versa, `conj` packing) against cgemm_gauss_kernel."""
    g = torch.Generator().manual_seed(9)
    dev = "cuda"
    keep, keep_log = ops.WINO, ops.LAUNCH_LOG
    try:
        for transposed, cin, cout, F in ((True, 24, 72, 9), (False, 128, 136, 17), (True, 136, 128, 6)):
            x = torch.randn(3, cin, F, 37, 2, generator=g)
            shape = (cin, cout, 5, 2) if transposed else (cout, cin, 5, 2)
            wr, wi = (torch.randn(shape, generator=g) * 0.1).to(dev), (torch.randn(shape, generator=g) * 0.1).to(dev)
            br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
            xp = ops.Planar.from_tensor5(x.to(dev), 38)
            ops.WINO = True
            g3 = ops.pack_cconv_gauss(wr, wi, br, bi, None, transposed=transposed)
            res = {}
            for wino in (False, True):
                ops.WINO = wino
                ops.LAUNCH_LOG = []
                st = torch.zeros(cout, 5, dtype=torch.float64, device=dev)
                y = ops.cconv2d(xp, None, None, cout, transposed=transposed, stats=st, gauss=g3)
                assert bool([c for c, *_ in ops.LAUNCH_LOG if c >= ops.WINO_CFG]) == wino
                res[wino] = (y.tensor5().cpu(), st.cpu())
            assert relerr(res[True][0], res[False][0]) < 5e-6
            assert relerr(res[True][1], res[False][1]) < 1e-5
            # adjoint: dy has the OUTPUT geometry; the data gradient is the other operator with conjugate-transposed weights
            dy = ops.Planar.from_tensor5(torch.randn(3, cout, y.F, 37, 2, generator=g).to(dev), 38)
            ops.WINO = True
            ga = ops.pack_cconv_gauss(wr, wi, None, None, None, adjoint_of=(cin, cout, cout, not transposed))
            d = {}
            for wino in (False, True):
                ops.WINO = wino
                ops.LAUNCH_LOG = []
                d[wino] = ops.cconv_dgrad(dy, None, None, cin, transposed, True, gauss=ga).tensor5().cpu()
                served = bool(ops.L.lib().idv_cconv_wino_supported(int(not transposed), cout, 0, cin, dy.F))
                assert bool([c for c, *_ in ops.LAUNCH_LOG if c >= ops.WINO_CFG]) == (wino and served)
            assert relerr(d[True], d[False]) < 5e-6
    finally:
        ops.WINO, ops.LAUNCH_LOG = keep, keep_log


@pytest.mark.parametrize("ns,B0,c0,c1,cout,F,T", [(2, 3, 8, 8, 12, 9, 30), (5, 2, 16, 16, 40, 17, 21), (3, 1, 4, 6, 4, 33, 45)])
def test_cconv_gauss_skip_half_once(ops, ns, B0, c0, c1, cout, F, T):
    """Repeated skips (pvae_module.py:2563-2567): the skip half of the transposed conv computed once per utterance and added in
    the epilogue (idv_cconv2d_gauss_fwd addend) against the oracle on the materialised repeat, with fold + PReLU."""
    g = torch.Generator().manual_seed(31 + ns)
    dev = "cuda"
    x = torch.randn(B0 * ns, c0, F, T, 2, generator=g)
    sk = torch.randn(B0, c1, F, T, 2, generator=g)
    wr, wi = torch.randn(c0 + c1, cout, 5, 2, generator=g) * 0.2, torch.randn(c0 + c1, cout, 5, 2, generator=g) * 0.2
    br, bi = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    want = O.complex_conv_transpose2d(torch.cat([x, sk.repeat_interleave(ns, dim=0)], dim=1), wr, br, wi, bi, (2, 1), (2, 0), True)
    mom = torch.stack([torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, 0.5 + torch.rand(cout, generator=g),
                       0.1 * torch.randn(cout, generator=g), 0.5 + torch.rand(cout, generator=g)])
    gam = [1 + 0.1 * torch.randn(cout, generator=g), torch.randn(cout, generator=g), 1 + 0.1 * torch.randn(cout, generator=g)]
    bet = [0.1 * torch.randn(cout, generator=g), 0.1 * torch.randn(cout, generator=g)]
    want = O.prelu(O.cbn_whiten_affine(want, mom[0], mom[1], mom[2], mom[3], mom[4], gam[0], gam[1], gam[2], bet[0], bet[1]),
                   torch.tensor(0.2))
    fold = ops.cbn_fold(mom.to(dev), *[t.to(dev) for t in gam], *[t.to(dev) for t in bet])
    slope = torch.tensor([0.2], device=dev)
    xp, skp = ops.Planar.from_tensor5(x.to(dev), T + 1), ops.Planar.from_tensor5(sk.to(dev), T + 1)
    wr_d, wi_d = wr.to(dev), wi.to(dev)
    g_skip = ops.pack_cconv_gauss_skip_part(wr_d, wi_d, c0)
    y_skip = ops.cconv2d(skp, None, None, cout, transposed=True, gauss=g_skip)
    g_main = ops.pack_cconv_gauss(wr_d, wi_d, br.to(dev), bi.to(dev), fold, cin_used=c0, transposed=True)
    y = ops.cconv2d(xp, None, None, cout, transposed=True, slope=slope, gauss=g_main, addend=y_skip, addend_div=ns)
    assert relerr(y.tensor5().cpu(), want) < TOL
    assert float(y.planes()[..., 0].abs().max()) == 0.0


def test_cconv_fold_prelu(ops):
    _conv_case(ops, True, False, 8, 16, 33, 21, 2, seed=2, fold=True, slope=0.2)
    _conv_case(ops, True, True, 8, 16, 9, 21, 2, seed=3, fold=True, slope=0.3)


def test_cconvt_skip_concat(ops):
    _conv_case(ops, True, True, 8, 4, 9, 30, 2, seed=4, skip_c=8)
    _conv_case(ops, True, True, 8, 4, 9, 30, 6, seed=5, skip_c=8, skip_div=3)       # repeated skips (num_samples=3)
    _conv_case(ops, True, True, 16, 1, 17, 30, 4, seed=6, skip_c=16, skip_div=2, fold=True, slope=0.25)


@pytest.mark.parametrize("name,transposed,causal", [("op_cconv_causal", False, True), ("op_cconv_plain", False, False),
                                                      ("op_cconvt_causal", True, True), ("op_cconvt_plain", True, False)])
def test_cconv_golden(ops, golden, name, transposed, causal):
    d = golden(name)
    x, want = T_(d["x"]), T_(d["y"])
    cin, cout, seed = int(d["cin"]), int(d["cout"]), int(d["seed"])
    pre = "tconv" if transposed else "conv"
    shape = (cin, cout, 5, 2) if transposed else (cout, cin, 5, 2)
    w = {k: O.synth_tensor(f"{pre}_{k}.weight", shape, seed).cuda() for k in ("re", "im")}
    b = {k: O.synth_tensor(f"{pre}_{k}.bias", (cout,), seed).cuda() for k in ("re", "im")}
    xp = ops.Planar.from_tensor5(x.cuda(), max(x.shape[3], want.shape[3]) + 1)
    wfrag, bias = ops.pack_cconv(w["re"], w["im"], b["re"], b["im"], None, transposed=transposed)
    y = ops.cconv2d(xp, wfrag, bias, cout, transposed=transposed, causal=causal)
    assert relerr(y.tensor5().cpu(), want) < TOL


def test_cbn_train_and_running(ops, golden):
    """Train-mode conv block: conv epilogue moments -> finalize -> apply+PReLU, and the running buffers
    (first call copies, second call 0.9/0.1 blend) against the reference's buffers."""
    d = golden("op_cbn")
    C, seed = int(d["C"]), int(d["seed"])
    x1, x2 = T_(d["x"]), T_(d["x2"])
    dev = "cuda"

    class BN:
        pass
    bn = BN()
    for k in ("gamma_rr", "gamma_ri", "gamma_ii", "beta_r", "beta_i"):
        setattr(bn, k, O.synth_tensor(k, (C,), seed).to(dev))
    for k in ("running_mean_real", "running_mean_imag", "Vrr", "Vri", "Vii"):
        setattr(bn, k, O.synth_tensor(k, (1, C, 1, 1), seed).to(dev).contiguous())
    # an identity "conv" that yields x itself is not available; run the stats through a real conv instead:
    g = torch.Generator().manual_seed(7)
    cin = 4
    wr, wi = torch.randn(C, cin, 5, 2, generator=g) * 0.3, torch.randn(C, cin, 5, 2, generator=g) * 0.3
    br, bi = torch.randn(C, generator=g), torch.randn(C, generator=g)
    wfrag, bias = ops.pack_cconv(wr.to(dev), wi.to(dev), br.to(dev), bi.to(dev), None)
    slope = torch.tensor([0.25], device=dev)
    run = {}
    for call_idx, seed_x in enumerate((11, 12)):
        gx = torch.Generator().manual_seed(seed_x)
        x = torch.randn(3, cin, 17, 13, 2, generator=gx) * (1.0 + call_idx) + 0.2
        raw = O.complex_conv2d(x, wr, br, wi, bi, (2, 1), (2, 1), True)
        st = O.cbn_batch_stats(raw)
        want = O.prelu(O.cbn_whiten_affine(raw, *st, bn.gamma_rr.cpu(), bn.gamma_ri.cpu(), bn.gamma_ii.cpu(),
                                           bn.beta_r.cpu(), bn.beta_i.cpu()), torch.tensor(0.25))
        xp = ops.Planar.from_tensor5(x.to(dev), 15)
        stats = torch.zeros(C, 5, dtype=torch.float64, device=dev)
        y = ops.cconv2d(xp, wfrag, bias, C, stats=stats)
        mom = ops.cbn_train(y, stats, bn, slope, first_call=(call_idx == 0))
        assert relerr(y.tensor5().cpu(), want) < 5e-5
        for k, s in zip(range(5), st):
            assert relerr(mom[k].cpu(), s.flatten()) < 5e-5
        run[call_idx] = [s.flatten().clone() for s in st]
        assert float(y.planes()[..., 0].abs().max()) == 0.0
    for k, name in enumerate(("running_mean_real", "running_mean_imag", "Vrr", "Vri", "Vii")):
        want = 0.9 * run[0][k] + 0.1 * run[1][k]
        assert relerr(getattr(bn, name).flatten().cpu(), want) < 5e-5
    # eval-mode fold against the reference's own eval output
    fold = ops.cbn_fold(torch.stack([T_(d["first_" + n]).flatten() for n in
                                     ("running_mean_real", "running_mean_imag", "Vrr", "Vri", "Vii")]).to(dev),
                        bn.gamma_rr, bn.gamma_ri, bn.gamma_ii, bn.beta_r, bn.beta_i)
    xp = ops.Planar.from_tensor5(x1.to(dev))
    import importlib
    L = importlib.import_module("i-dccrn-vae_amd")._lib
    L.call("idv_cbn_apply_prelu", xp.ptr(), L.p(fold), L.p(None), L.i(C), L.i(xp.F), L.i(xp.B), L.i(xp.Tp), L.i(xp.Jp),
           L.i(xp.T), L.stream_ptr())
    st1 = O.cbn_batch_stats(x1)
    want = O.cbn_whiten_affine(x1, *st1, bn.gamma_rr.cpu(), bn.gamma_ri.cpu(), bn.gamma_ii.cpu(), bn.beta_r.cpu(),
                               bn.beta_i.cpu())
    assert relerr(xp.tensor5().cpu(), want) < TOL
    assert relerr(xp.tensor5().cpu(), T_(d["y_train"])) < TOL


def test_stft_istft_golden(ops, golden):
    d = golden("op_stft")
    x = T_(d["x"]).cuda()
    plan = ops.DftPlan(NFFT, WIN, HOP, 1 + x.shape[1] // HOP, "cuda")
    X = ops.stft(x, plan)
    assert relerr(X.tensor4().cpu(), T_(d["X"])) < TOL
    assert X.T == 1 + x.shape[1] // HOP                                    # frame indexing is exact
    Y = T_(d["Y"])                                                        # [B,F,T,2]
    Yp = ops.Planar.from_tensor5(Y.unsqueeze(1).cuda())
    y = ops.istft(Yp, plan)
    assert y.shape == tuple(d["y"].shape)                                 # length hop*(T-1)
    assert relerr(y.cpu(), T_(d["y"])) < TOL


@pytest.mark.parametrize("L", [1600, 3201, 6400])
def test_stft_roundtrip_and_oracle(ops, L):
    g = torch.Generator().manual_seed(L)
    x = torch.randn(3, L, generator=g) * 0.1
    plan = ops.DftPlan(NFFT, WIN, HOP, 1 + L // HOP, "cuda")
    X = ops.stft(x.cuda(), plan)
    assert relerr(X.tensor4().cpu(), O.stft(x, NFFT, HOP, WIN)) < TOL
    y = ops.istft(X, plan).cpu()
    n = y.shape[1]
    assert n == HOP * (L // HOP)
    assert relerr(y, x[:, :n]) < 1e-5                                      # STFT -> ISTFT is the identity


@pytest.mark.parametrize("H,I,T,B", [(16, 20, 7, 3), (128, 64, 12, 5), (128, 160, 9, 18), (48, 32, 6, 2), (96, 160, 5, 17)])
def test_clstm(ops, golden, H, I, T, B):
    if (H, I, T, B) == (16, 20, 7, 3):
        d = golden("op_clstm")
        x, want, seed = T_(d["x"]), T_(d["y"]), int(d["seed"])
    else:
        g = torch.Generator().manual_seed(H + T)
        x, seed = torch.randn(T, B, I, 2, generator=g), 77
        want = None
    names = [f"lstm_{s}.{w}_l{l}" for s in ("re", "im") for l in (0, 1) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    sd = {}
    for n in names:
        l = int(n[-1])
        if "weight_ih" in n:
            shape = (4 * H, I if l == 0 else H)
        elif "weight_hh" in n:
            shape = (4 * H, H)
        else:
            shape = (4 * H,)
        sd[n] = O.synth_tensor(n, shape, seed) * (3.0 if "weight" in n else 1.0)
    if want is None:
        want = O.complex_lstm(x, sd, "", 2)
    else:
        sd = {n: O.synth_tensor(n, tuple(sd[n].shape), seed) for n in names}
    # planar input: [T,B,I,2] -> reference layout [B, C=I, F=1, T, 2]
    xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).cuda())
    get = lambda n: sd[n].cuda()
    p0 = ops.pack_lstm(get, H, I, 0, "cuda")
    p1 = ops.pack_lstm(get, H, H, 1, "cuda")
    out = ops.clstm(xp, p0, p1, H)
    got = out.channel_slice(0, H).cpu().permute(1, 0, 2, 3)                # [T,B,H,2]
    assert relerr(got, want) < TOL
    assert float(out.planes()[..., 0].abs().max()) == 0.0


_LSTM_ORACLE = {}


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("H,I,T,B", [(384, 1280, 641, 2), (768, 1280, 641, 3), (768, 1280, 40, 18)])
def test_clstm_vae_sizes(ops, precision, H, I, T, B):
    """The hidden sizes the VAE encoders use (3*zdim = 384, 6*zdim = 768 at zdim 128; BASELINE configs 2-5) at the real
    input width and a full 4 s sequence, against the oracle in float64: the per-step recurrent kernel, its > 64 KB
    dynamic-LDS branch (H = 768) and the planar layer-1 projection."""
    g = torch.Generator().manual_seed(H + T)
    x = torch.randn(T, B, I, 2, generator=g) * 0.5
    names = [f"lstm_{s}.{w}_l{l}" for s in ("re", "im") for l in (0, 1) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    sd = {}
    for n in names:
        l = int(n[-1])
        shape = (4 * H, I if l == 0 else H) if "weight_ih" in n else ((4 * H, H) if "weight_hh" in n else (4 * H,))
        sd[n] = O.synth_tensor(n, shape, 91) * (2.0 if "weight" in n else 1.0)
    key = (H, I, T, B)
    if key not in _LSTM_ORACLE:                   # the float64 oracle walks T steps in Python: once per shape, not per mode
        _LSTM_ORACLE[key] = O.complex_lstm(x.double(), {k: v.double() for k, v in sd.items()}, "", 2)
    want = _LSTM_ORACLE[key]
    xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).cuda())
    get = lambda n: sd[n].cuda()
    keep = ops.PRECISION
    try:
        ops.set_precision(precision)
        p0 = ops.pack_lstm(get, H, I, 0, "cuda")
        p1 = ops.pack_lstm(get, H, H, 1, "cuda")
        out = ops.clstm(xp, p0, p1, H)
        got = out.channel_slice(0, H).cpu().permute(1, 0, 2, 3)
    finally:
        ops.set_precision(keep)
    assert torch.isfinite(got).all()
    assert relerr(got, want) < (TOL if precision == "fp32" else 2e-4)
    assert float(out.planes()[..., 0].abs().max()) == 0.0


@pytest.mark.parametrize("H,B,T", [(384, 5, 60), (768, 32, 48), (384, 40, 30), (768, 33, 25), (768, 64, 20), (384, 64, 33)])
def test_clstm_persistent_f32_matches_per_step(ops, H, B, T):
    """The exact-fp32 persistent cooperative recurrence (lstm_pers_f32.hip) against the per-step fp32 kernels it replaces:
    1, 2 and 4 row tiles per workgroup, several batch chunks, a ragged last tile; same fp32 products in another summation
    order -> 2e-6; bit-repeatable."""
    g = torch.Generator().manual_seed(H + B + 1)
    I = 64
    x = torch.randn(T, B, I, 2, generator=g) * 0.5
    names = [f"lstm_{s}.{w}_l{l}" for s in ("re", "im") for l in (0, 1) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    sd = {}
    for n in names:
        l = int(n[-1])
        shape = (4 * H, I if l == 0 else H) if "weight_ih" in n else ((4 * H, H) if "weight_hh" in n else (4 * H,))
        sd[n] = O.synth_tensor(n, shape, 93) * (2.0 if "weight" in n else 1.0)
    xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).cuda())
    get = lambda n: sd[n].cuda()
    keep = (ops.PRECISION, ops.LSTM_PERSISTENT)
    try:
        ops.set_precision("fp32")
        p0, p1 = ops.pack_lstm(get, H, I, 0, "cuda"), ops.pack_lstm(get, H, H, 1, "cuda")
        ops.LSTM_PERSISTENT = False
        ref = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        ops.LSTM_PERSISTENT = True
        assert amd_lib().idv_lstm_pers_f32_supported(H, B)
        got = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        got2 = ops.clstm(xp, p0, p1, H).channel_slice(0, H)
        torch.cuda.synchronize()
        assert amd_lib().idv_coop_last_status(1) == 0
    finally:
        ops.set_precision(keep[0])
        ops.LSTM_PERSISTENT = keep[1]
    assert torch.isfinite(got).all()
    assert torch.equal(got, got2)
    assert relerr(got.cpu(), ref.cpu()) < 2e-6


@pytest.mark.parametrize("H,B,T", [(384, 5, 60), (768, 32, 48), (384, 40, 30), (768, 33, 25)])
def test_clstm_persistent_matches_per_step(ops, H, B, T):
    """The persistent cooperative recurrence (lstm_pers.hip: W_hh slices resident in registers, per-step arrive counter)
    against the per-step kernels on the same split-bf16 inputs; batch sizes that need 1, 2 and 3 batch chunks and a ragged
    last tile.  Different summation order and bf16x3 vs fp32 recurrent products -> 2e-4."""
    g = torch.Generator().manual_seed(H + B)
    I = 64
    x = torch.randn(T, B, I, 2, generator=g) * 0.5
    names = [f"lstm_{s}.{w}_l{l}" for s in ("re", "im") for l in (0, 1) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    sd = {}
    for n in names:
        l = int(n[-1])
        shape = (4 * H, I if l == 0 else H) if "weight_ih" in n else ((4 * H, H) if "weight_hh" in n else (4 * H,))
        sd[n] = O.synth_tensor(n, shape, 92) * (2.0 if "weight" in n else 1.0)
    xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).cuda())
    get = lambda n: sd[n].cuda()
    keep = (ops.PRECISION, ops.LSTM_PERSISTENT)
    try:
        ops.set_precision("bf16x3")
        p0, p1 = ops.pack_lstm(get, H, I, 0, "cuda"), ops.pack_lstm(get, H, H, 1, "cuda")
        ops.LSTM_PERSISTENT = False
        ref = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        ops.LSTM_PERSISTENT = True
        assert amd_lib().idv_lstm_pers_supported(H, B)
        got = ops.clstm(xp, p0, p1, H).channel_slice(0, H)
        got2 = ops.clstm(xp, p0, p1, H).channel_slice(0, H)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(keep[0])
        ops.LSTM_PERSISTENT = keep[1]
    assert torch.isfinite(got).all()
    assert torch.equal(got, got2)                               # deterministic (fixed reduction order)
    assert relerr(got.cpu(), ref.cpu()) < 2e-4


@pytest.mark.parametrize("B,T", [(5, 40), (40, 33), (64, 641), (100, 7)])
def test_clstm_coop_f32_matches_one_cu(ops, B, T):
    """The exact-fp32 H = 128 recurrence spread over four CUs per sequence tile (lstm_coop_f32.hip: gate columns split over
    workgroups, h exchanged through global memory) against the one-CU register-resident kernel on the same inputs: same
    products, another summation order -> 1e-6; ragged last tile; bit-repeatable; the full 641-frame length."""
    H, I = 128, 64
    g = torch.Generator().manual_seed(B)
    x = torch.randn(T, B, I, 2, generator=g) * 0.5
    names = [f"lstm_{s}.{w}_l{l}" for s in ("re", "im") for l in (0, 1) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    sd = {}
    for n in names:
        l = int(n[-1])
        shape = (4 * H, I if l == 0 else H) if "weight_ih" in n else ((4 * H, H) if "weight_hh" in n else (4 * H,))
        sd[n] = O.synth_tensor(n, shape, 93) * (2.0 if "weight" in n else 1.0)
    xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).cuda())
    get = lambda n: sd[n].cuda()
    keep = (ops.PRECISION, ops.LSTM_PERSISTENT, ops.LSTM_STACK2)
    try:
        ops.set_precision("fp32")
        p0, p1 = ops.pack_lstm(get, H, I, 0, "cuda"), ops.pack_lstm(get, H, H, 1, "cuda")
        ops.LSTM_PERSISTENT = False                                  # flags bit 3: the one-CU kernel
        ref = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        ops.LSTM_PERSISTENT = True
        assert amd_lib().idv_lstm_coop_f32_supported(H, B)
        ops.LSTM_STACK2 = False                                      # one cooperative launch per layer
        got = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        got2 = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        # both layers in ONE cooperative launch, layer 1 a step behind layer 0 and its input projection inside the recurrence
        # (lstm_stack2_f32.hip): W_ih h0[t] is summed in another order than the hoisted GEMM does -> 1e-6, bit-repeatable
        ops.LSTM_STACK2 = True
        assert amd_lib().idv_lstm_stack2_f32_supported(H, B) and p1[4] is not None
        st = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        st2 = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        torch.cuda.synchronize()
    finally:
        ops.set_precision(keep[0])
        ops.LSTM_PERSISTENT = keep[1]
        ops.LSTM_STACK2 = keep[2]
    assert torch.isfinite(got).all() and torch.isfinite(st).all()
    assert torch.equal(got, got2) and torch.equal(st, st2)
    assert relerr(got.cpu(), ref.cpu()) < 1e-6
    assert relerr(st.cpu(), ref.cpu()) < 1e-6


@pytest.mark.parametrize("H,precision", [(128, "fp32"), (384, "bf16x3"), (384, "fp32")])
def test_cooperative_recurrence_times_out_instead_of_hanging(ops, H, precision):
    """Safety net of the cooperative recurrences (lstm_coop_f32.hip, lstm_pers.hip): with one workgroup withheld
    (IDV_COOP_FAULT=1 makes workgroup (0, 0, 0) return before its first arrive) every spin runs into its bound, the abort
    flag drains the grid, the outputs are poisoned with NaN -- the call returns within a second, nothing hangs, nothing
    looks like a result -- and the next call without the fault is correct again."""
    import os
    import time
    B, T, I = 5, 12, 64
    g = torch.Generator().manual_seed(H)
    x = torch.randn(T, B, I, 2, generator=g) * 0.5
    names = [f"lstm_{s}.{w}_l{l}" for s in ("re", "im") for l in (0, 1) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    sd = {}
    for n in names:
        l = int(n[-1])
        shape = (4 * H, I if l == 0 else H) if "weight_ih" in n else ((4 * H, H) if "weight_hh" in n else (4 * H,))
        sd[n] = O.synth_tensor(n, shape, 94)
    xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).cuda())
    get = lambda n: sd[n].cuda()
    keep = ops.PRECISION
    try:
        ops.set_precision(precision)
        p0, p1 = ops.pack_lstm(get, H, I, 0, "cuda"), ops.pack_lstm(get, H, H, 1, "cuda")
        good = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        torch.cuda.synchronize()
        os.environ["IDV_COOP_FAULT"] = "1"
        t0 = time.perf_counter()
        bad = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        os.environ.pop("IDV_COOP_FAULT")
        # the time-out is surfaced through the C ABI, not only through NaNs: the sticky status says -3 (IDV_ECOOP) ...
        lib = ops.L.lib()
        assert lib.idv_coop_last_status(0) == -3
        # ... and it stays, like a device error, until it is acknowledged: every cooperative entry refuses, none consumes it
        for _ in range(2):
            with pytest.raises(ops.L.IdvError, match="status -3"):
                ops.clstm(xp, p0, p1, H)
            assert lib.idv_coop_last_status(0) == -3
        # the owning operation's check (inference.py / bench.py / smoke call it after their forward) raises and acknowledges
        with pytest.raises(ops.CoopTimeout):
            ops.coop_check()
        assert lib.idv_coop_last_status(0) == 0 and not ops.coop_clear()
        again = ops.clstm(xp, p0, p1, H).channel_slice(0, H).clone()
        ops.coop_check()
        assert torch.equal(again, good) or torch.allclose(again, good, rtol=0, atol=1e-6)
    finally:
        os.environ.pop("IDV_COOP_FAULT", None)
        ops.set_precision(keep)
        torch.cuda.synchronize()
        ops.L.lib().idv_coop_last_status(1)              # never leave a sticky status behind for an unrelated test
    assert dt < 5.0, dt                                   # two layers x one 0.4 s spin bound
    assert torch.isnan(bad).any()
    assert torch.isfinite(good).all() and torch.equal(good, again)


def test_image_writing_variants_are_bit_identical(ops):
    """The fused forms added for the bf16x3 training mode against the two-pass forms they replace, bit for bit:
    planar -> image with the batch repeat inside (idv_planar_to_image_repeat), the out-of-place BN apply + PReLU and the BN
    backward apply that also write the split image of their result (idv_cbn_apply_prelu_to_img, idv_cbn_bwd_apply_img)."""
    g = torch.Generator().manual_seed(3)
    C, F, B, T = 8, 5, 3, 21
    x = ops.Planar.from_tensor5(torch.randn(B, C, F, T, 2, generator=g).cuda())
    # repeat inside the conversion
    a = ops.to_image_repeat(x, 2)
    b = ops.to_image(ops.repeat_batch(x, 2))
    n = (2 * C + 7) // 8 * F * a.Jp * 8
    off = ops.IMG_SLACK
    assert a.Jp == b.Jp and torch.equal(a.buf[off:off + n], b.buf[off:off + n])
    assert torch.equal(a.buf[off + a.lo_off:off + a.lo_off + n], b.buf[off + b.lo_off:off + b.lo_off + n])
    # BN apply + PReLU
    fold = (torch.randn(C, 6, generator=g) * 0.5).cuda()
    slope = torch.tensor([0.25], device="cuda")
    z0 = ops.cbn_apply_to(x, fold, slope)
    z1, zi = ops.cbn_apply_to(x, fold, slope, want_image=True)
    ref = ops.to_image(z0)
    assert torch.equal(z0.planes(), z1.planes())
    m = (2 * C + 7) // 8 * F * zi.Jp * 8
    assert torch.equal(zi.buf[off:off + m], ref.buf[off:off + m])
    assert torch.equal(zi.buf[off + zi.lo_off:off + zi.lo_off + m], ref.buf[off + ref.lo_off:off + ref.lo_off + m])
    # BN backward apply
    dz = ops.Planar.from_tensor5(torch.randn(B, C, F, T, 2, generator=g).cuda())
    moments = torch.stack([torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1, 0.5 + torch.rand(C, generator=g),
                           0.1 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)]).cuda()
    gam = [(1 + 0.1 * torch.randn(C, generator=g)).cuda(), torch.randn(C, generator=g).cuda(), (1 + 0.1 * torch.randn(C, generator=g)).cuda()]
    cnt = float(B * F * T)
    r0 = ops.cbn_bwd(dz, x, fold, moments, gam, slope, cnt)
    r1 = ops.cbn_bwd(dz, x, fold, moments, gam, slope, cnt, want_image=True)
    dy0, (dy1, dyi) = r0[0], r1[0]
    # the channel sums behind both calls are double-precision atomics (order-dependent in the last bits): 1e-6 between the
    # calls, bit-exact between the planar result and the image of ONE call
    assert relerr(dy1.planes().cpu(), dy0.planes().cpu()) < 1e-6
    ref = ops.to_image(dy1)
    assert torch.equal(dyi.buf[off:off + m], ref.buf[off:off + m])
    assert torch.equal(dyi.buf[off + dyi.lo_off:off + dyi.lo_off + m], ref.buf[off + ref.lo_off:off + ref.lo_off + m])
    for u, v in zip(r0[1:], r1[1:]):
        assert relerr(v.cpu(), u.cpu()) < 1e-6


def test_pw_bf16x3_rows_matches_the_fp32_swap_store(ops):
    """idv_pw_bf16x3_rows (dh = W^T dG in BPTT's row-major form, bf16x3 training) against idv_pw_gemm with the transposed
    store, same operands: 2e-5."""
    g = torch.Generator().manual_seed(4)
    K, M, B, T = 256, 96, 3, 37
    w = (torch.randn(M, K, generator=g) * 0.1).cuda()
    xp = ops.Planar.from_tensor5(torch.randn(B, K // 2, 1, T, 2, generator=g).cuda())        # K planes of Jp columns
    wf = ops.pack_pw(w, None)
    want = torch.zeros(T * B, M, device="cuda")
    ops.pw_gemm(xp.ptr(), K, wf[0], wf[1], M, B, xp.Tp, xp.Jp, T, ops.L._P(want.data_ptr()), swap=True, ldo=M)
    kimg = ops.KImage.from_planes(xp.ptr(), K, B * xp.Tp, xp.Jp, "cuda", pad_to=64)
    got = torch.zeros(T * B, M, device="cuda")
    ops.pw_bf16x3_rows(kimg, K, ops.pack_pw_bf16(w), torch.zeros(M, device="cuda"), M, M, B, T, xp.Tp, ops.L._P(got.data_ptr()))
    torch.cuda.synchronize()
    assert relerr(got.cpu(), want.cpu()) < 2e-5


def amd_lib():
    import importlib
    return importlib.import_module("i-dccrn-vae_amd")._lib.lib()


def test_cdense_golden(ops, golden):
    d = golden("op_cdense")
    x, want, seed = T_(d["x"]), T_(d["y"]), int(d["seed"])              # x: [21,16,2]
    wr, br = O.synth_tensor("linear_read.weight", (40, 16), seed), O.synth_tensor("linear_read.bias", (40,), seed)
    wi, bi = O.synth_tensor("linear_imag.weight", (40, 16), seed), O.synth_tensor("linear_imag.bias", (40,), seed)
    # rows of x are (b*T + t): B=3, T=7
    xr = x.reshape(3, 7, 16, 2).permute(0, 2, 1, 3).unsqueeze(2)         # [B, C=16, 1, T, 2]
    xp = ops.Planar.from_tensor5(xr.cuda())
    out = ops.cdense(xp, ops.pack_pw(wr.cuda(), br.cuda()), ops.pack_pw(wi.cuda(), bi.cuda()), 40, 8, 5)
    got = out.tensor5().cpu()                                            # [B, 8, 5, T, 2]
    got = got.permute(0, 3, 1, 2, 4).reshape(21, 40, 2)
    assert relerr(got, want) < TOL


def test_mask_apply(ops):
    g = torch.Generator().manual_seed(3)
    B, F, T = 4, 257, 19
    M = torch.randn(B, F, T, 2, generator=g)
    M[0, 0, 0] = 0.0                                                      # |M| = 0 -> atan2(0,0) = 0 branch
    X = torch.randn(B // 2, F, T, 2, generator=g)
    want = O.apply_mask(M, X.repeat_interleave(2, dim=0))
    Mp = ops.Planar.from_tensor5(M.unsqueeze(1).cuda())
    Xp = ops.Planar.from_tensor5(X.unsqueeze(1).cuda())
    pred, pc = ops.mask_apply(Mp, Xp, x_div=2)
    assert relerr(pred.tensor4().cpu(), want) < TOL
    assert relerr(torch.view_as_real(pc).cpu(), want) < TOL
    assert relerr(torch.view_as_real(ops.planar_to_complex(pred)).cpu(), want) < TOL


def _lat(ops, miu, ls, dl):
    """[B,T,H,2] x3 -> planar latent with channels [miu | log_sigma | delta]."""
    lat = torch.cat([miu, ls, dl], dim=2)                                 # [B,T,3H,2]
    return ops.Planar.from_tensor5(lat.permute(0, 2, 1, 3).unsqueeze(2).cuda())


def test_reparam_and_kl_golden(ops, golden):
    d = golden("op_vae")
    miu, ls, dl = T_(d["miu"]), T_(d["ls"]), T_(d["dl"])
    H = miu.shape[2]
    lat = _lat(ops, miu, ls, dl)
    ns = d["eps_r"].shape[1]
    z = ops.reparam(lat, (0, H, 2 * H), H, T_(d["eps_r"]).cuda(), T_(d["eps_i"]).cuda(), ns)
    got = z.channel_slice(0, H).cpu()                                      # [B*ns, T, H, 2]
    assert relerr(got, T_(d["z"])) < TOL
    lat2 = _lat(ops, T_(d["miu2"]), T_(d["ls2"]), T_(d["dl2"]))
    off = (0, H, 2 * H)
    assert relerr(ops.ckl(lat, off, lat2, off, H, 1e-9).cpu(), T_(d["kl_pretrain"])) < TOL
    assert relerr(ops.ckl(lat, off, lat2, off, H, 1e-10).cpu(), T_(d["kl_nsvae"]).mean()) < TOL
    prior = O.complex_kl(miu, torch.zeros_like(miu), ls, torch.zeros_like(ls), dl, torch.zeros_like(dl), 1e-9).mean()
    assert relerr(ops.ckl(lat, off, None, None, H, 1e-9).cpu(), prior) < TOL
    lat3 = _lat(ops, T_(d["miu3"]), T_(d["ls3"]), T_(d["dl3"]))
    want = torch.sqrt(torch.sum(torch.mean((miu - T_(d["miu3"])) ** 2, dim=(0, 1))))
    assert relerr(ops.miu_dist(lat, 0, lat3, 0, H).cpu(), want) < TOL
    assert relerr(ops.miu_dist(lat, 0, lat3, 0, H).cpu(), T_(d["nsvae"])[4]) < TOL


def test_guard_branch_forced(ops):
    """|delta| >= sigma - 1e-3 must take the rescale branch (rare on random data): force it."""
    B, T, H, ns = 2, 3, 4, 2
    g = torch.Generator().manual_seed(9)
    miu = torch.randn(B, T, H, 2, generator=g)
    ls = torch.full((B, T, H, 2), -1.0)
    dl = torch.randn(B, T, H, 2, generator=g) * 3.0                        # |delta| >> sigma = e^-1
    er, ei = torch.randn(B, ns, T, H, generator=g), torch.randn(B, ns, T, H, generator=g)
    lat = _lat(ops, miu, ls, dl)
    z = ops.reparam(lat, (0, H, 2 * H), H, er.cuda(), ei.cuda(), ns)
    assert relerr(z.channel_slice(0, H).cpu(), O.reparameterization(miu, ls, dl, ns, er, ei)) < TOL
    want = O.complex_kl(miu, torch.zeros_like(miu), ls, torch.zeros_like(ls), dl, torch.zeros_like(dl), 1e-9).mean()
    assert relerr(ops.ckl(lat, (0, H, 2 * H), None, None, H, 1e-9).cpu(), want) < 1e-4


def test_sisnr_and_recon_golden(ops, golden):
    d = golden("op_sisnr")
    got = ops.sisnr(T_(d["src"]).cuda(), T_(d["est"]).cuda())
    assert abs(float(got) - (-24.9485)) < 1e-3                             # model/sisnr_loss.py:27-30 known answer
    assert relerr(got.cpu(), T_(d["known"])) < TOL
    assert relerr(ops.sisnr(T_(d["s2"]).cuda(), T_(d["e2"]).cuda()).cpu(), T_(d["r2"])) < TOL
    s, e = T_(d["s2"]), T_(d["e2"])
    want = O.si_snr(s.repeat_interleave(2, dim=0), torch.cat([e, e * 0.5 + 0.01], 0)[[0, 4, 1, 5, 2, 6, 3, 7]])
    est = torch.cat([e, e * 0.5 + 0.01], 0)[[0, 4, 1, 5, 2, 6, 3, 7]].contiguous()
    assert relerr(ops.sisnr(s.cuda(), est.cuda(), src_div=2).cpu(), want) < TOL
    r = golden("op_recon")
    P, Or = T_(r["P"]), T_(r["O"])
    pc = torch.view_as_complex(P.contiguous().cuda())
    cpx, mag = ops.recon_loss(pc, Or.cuda())
    assert relerr(cpx.cpu(), T_(r["want"])[1]) < TOL
    assert relerr(mag.cpu(), T_(r["want"])[2]) < TOL
    # strided "ori" (a planar view) gives the same result
    Op = ops.Planar.from_tensor5(Or.unsqueeze(1).cuda())
    cpx2, mag2 = ops.recon_loss(pc, Op.tensor4())
    assert relerr(cpx2.cpu(), T_(r["want"])[1]) < TOL and relerr(mag2.cpu(), T_(r["want"])[2]) < TOL


def test_invalid_arguments_are_reported(amd):
    L = amd._lib
    with pytest.raises(L.IdvError):
        L.call("idv_sisnr", L.p(None), L.i(1), L.i(1), L.p(None), L.i(1), L.i(1), L.i(1), L.p(None), L.p(None), L.stream_ptr())


# ----------------------------------------------------------------------------- split-precision (bf16x3) path
@pytest.mark.parametrize("causal,transposed,cin,cout,F,T,B,skip_c", [
    (True, False, 32, 64, 65, 70, 2, 0), (True, False, 16, 32, 17, 130, 3, 0), (False, False, 8, 32, 33, 40, 2, 0),
    (True, True, 16, 64, 9, 70, 2, 16), (True, True, 32, 32, 5, 140, 3, 0), (True, True, 64, 128, 17, 40, 2, 64),
    (True, False, 128, 128, 33, 70, 1, 0), (True, True, 8, 64, 33, 30, 2, 8),
])
def test_cconv_bf16x3(ops, causal, transposed, cin, cout, F, T, B, skip_c):
    """bf16x3 split-precision contraction vs the oracle: per-operator 2e-4 (product error ~2^-16), guard columns exact."""
    g = torch.Generator().manual_seed(11)
    dev = "cuda"
    cin_tot = cin + skip_c
    x = torch.randn(B, cin, F, T, 2, generator=g)
    shape = (cin_tot, cout, 5, 2) if transposed else (cout, cin_tot, 5, 2)
    wr, wi = torch.randn(shape, generator=g) * 0.1, torch.randn(shape, generator=g) * 0.1
    br, bi = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    xin, sk = x, None
    if skip_c:
        sk = torch.randn(B, skip_c, F, T, 2, generator=g)
        xin = torch.cat([x, sk], dim=1)
    if transposed:
        want = O.complex_conv_transpose2d(xin.double(), wr.double(), br.double(), wi.double(), bi.double(), (2, 1), (2, 0), causal)
    else:
        want = O.complex_conv2d(xin.double(), wr.double(), br.double(), wi.double(), bi.double(), (2, 1),
                                (2, 1) if causal else (2, 0), causal)
    C = cout
    mom = torch.stack([torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1, 0.5 + torch.rand(C, generator=g),
                       0.1 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)])
    gam = [1 + 0.1 * torch.randn(C, generator=g), torch.randn(C, generator=g), 1 + 0.1 * torch.randn(C, generator=g)]
    bet = [0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)]
    want = O.prelu(O.cbn_whiten_affine(want.float(), mom[0], mom[1], mom[2], mom[3], mom[4], gam[0], gam[1], gam[2], bet[0], bet[1]),
                   torch.tensor(0.2))
    fold = ops.cbn_fold(mom.to(dev), *[t.to(dev) for t in gam], *[t.to(dev) for t in bet])
    Tp = max(T, want.shape[3]) + 1
    xp = ops.Planar.from_tensor5(x.to(dev), Tp)
    skp = ops.Planar.from_tensor5(sk.to(dev), Tp) if sk is not None else None
    assert ops.bf16_supported(transposed, cin, skip_c, 1, cout)
    wfrag, bias = ops.pack_cconv(wr.to(dev), wi.to(dev), br.to(dev), bi.to(dev), fold, transposed=transposed)
    wbf = ops.pack_cconv_bf16(wr.to(dev), wi.to(dev), fold, transposed=transposed)
    slope = torch.tensor([0.2], device=dev)
    y = ops.cconv2d(xp, wfrag, bias, cout, transposed=transposed, causal=causal, slope=slope, skip=skp, wfrag_bf16=wbf)
    y32 = ops.cconv2d(xp, wfrag, bias, cout, transposed=transposed, causal=causal, slope=slope, skip=skp)
    got = y.tensor5().cpu()
    assert got.shape == want.shape
    e16, e32 = relerr(got, want), relerr(y32.tensor5().cpu(), want)
    assert e32 < TOL and e16 < 2e-4, (e16, e32)
    assert float(y.planes()[..., 0].abs().max()) == 0.0
    if y.planes().shape[-1] > y.T + 1:
        assert float(y.planes()[..., y.T + 1:].abs().max()) == 0.0
    # split-image form of the same launch: image sources give the identical planar result (same operand split),
    # the image output decodes to it within the split's 2^-16, and planar sources can write an image too
    if cout % 4 == 0:
        xi, ski = ops.to_image(xp), (ops.to_image(skp) if skp is not None else None)
        back = ops.to_planar(xi)
        assert float((back.tensor5() - xp.tensor5()).abs().max()) <= 2.0 ** -16 * float(xp.tensor5().abs().max())
        kw = dict(transposed=transposed, causal=causal, slope=slope)
        yp, yi = ops.cconv2d_img(xi, wbf, bias, cout, skip=ski, want_planar=True, want_image=True, **kw)
        assert torch.equal(yp.tensor5(), y.tensor5())
        dec = ops.to_planar(yi)
        assert float((dec.tensor5() - y.tensor5()).abs().max()) <= 2.0 ** -16 * float(y.tensor5().abs().max())
        assert float(dec.planes()[..., 0].abs().max()) == 0.0
        _, yi2 = ops.cconv2d_img(xp, wbf, bias, cout, skip=skp, want_planar=False, want_image=True, **kw)
        assert torch.equal(ops.to_planar(yi2).tensor5(), dec.tensor5())
        yp3, none = ops.cconv2d_img(xi, wbf, bias, cout, skip=ski, want_planar=True, want_image=False, **kw)
        assert none is None and torch.equal(yp3.tensor5(), y.tensor5())


def test_image_path_matches_planar_path(ops, amd):
    """standard_DCCRN eval on split images (default) vs planar fp32 inter-layer activations: identical arithmetic,
    so identical waveforms; full-width net (every block on the bf16x3 kernels) and a narrow one (fallbacks)."""
    import importlib
    pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
    keep = (ops.IMAGE_PATH, ops.PRECISION)
    try:
        ops.set_precision("bf16x3")
        for base, L_, B in ((32, 3000, 3), (32, 700, 1), (32, 16000, 5), (8, 2345, 2), (4, 1600, 2)):
            np_ = O.net_params(True, base)
            m = pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, [0, 1, 2, 3, 4, 5], "mask", False, None, None)
            m.load_state_dict(O.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 21))
            m = m.cuda()
            g = torch.Generator().manual_seed(base)
            x = (torch.randn(B, L_, generator=g) * 0.1).cuda()
            ops.IMAGE_PATH = True
            c1, p1 = m(x, train=False)
            lat1 = m.std_DCCRN.latent.clone()
            ops.IMAGE_PATH = False
            c0, p0 = m(x, train=False)
            if base == 32:
                assert torch.equal(c1, c0), (base, float((c1 - c0).abs().max()))
                assert torch.equal(torch.view_as_real(p1), torch.view_as_real(p0))
                assert torch.equal(lat1, m.std_DCCRN.latent)
            else:       # narrow blocks fall back to the fp32 kernel, which then reads an image decoded to hi + lo (2^-17)
                assert relerr(c1, c0) < 1e-5 and relerr(lat1, m.std_DCCRN.latent) < 1e-5
    finally:
        ops.IMAGE_PATH = keep[0]
        ops.set_precision(keep[1])


def test_bf16x3_train_stats_and_model(ops, amd, golden):
    """bf16x3 end to end: the full-size DCCRN-CL golden waveform within 1e-3 (north_star tolerance) and the
    train-mode moments of the split-precision conv."""
    import importlib
    pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
    d = golden("dccrn_full_eval")
    np_ = O.net_params(True, 32)
    m = pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, [0, 1, 2, 3, 4, 5], "mask", False, None, None)
    m.load_state_dict(O.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(d["seed"])))
    m = m.cuda()
    x = T_(d["x"]).cuda()
    try:
        ops.set_precision("bf16x3")
        clean, _ = m(x, train=False)
        e = relerr(clean.cpu(), T_(d["clean"]))
        assert e < 1e-3, e
        # reduced-width model in train mode (batch statistics from the split-precision epilogue)
        np4 = O.net_params(True, 8)
        m2 = pm.DCCRN_(NFFT, HOP, np4, True, "cuda", WIN, [0, 1, 2, 3, 4, 5], "mask", False, None, None)
        sd = O.synth_state_dict({k: tuple(v.shape) for k, v in m2.state_dict().items()}, 9)
        m2.load_state_dict(sd)
        m2 = m2.cuda()
        g = torch.Generator().manual_seed(4)
        xs = torch.randn(2, 1600, generator=g) * 0.1
        c16, _ = m2(xs.cuda(), train=True)
        ops.set_precision("fp32")
        m3 = pm.DCCRN_(NFFT, HOP, np4, True, "cuda", WIN, [0, 1, 2, 3, 4, 5], "mask", False, None, None)
        m3.load_state_dict(sd)
        m3 = m3.cuda()
        c32, _ = m3(xs.cuda(), train=True)
        assert relerr(c16.cpu(), c32.cpu()) < 1e-3
        for k, v in m3.state_dict().items():
            if k.endswith("Vrr"):
                assert relerr(m2.state_dict()[k].cpu(), v.cpu()) < 1e-3, k
    finally:
        ops.set_precision("fp32")
    print("bf16x3 full-size waveform rel err vs reference golden:", e)


@pytest.mark.parametrize("T,B,I", [(40, 5, 64), (641, 3, 160)])
def test_clstm_bf16x3(ops, T, B, I):
    """Split-bf16 recurrence (H = 128) against the fp64 oracle: 641 recurrent steps must not accumulate error."""
    H = 128
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, B, I, 2, generator=g)
    names = [f"lstm_{s}.{w}_l{l}" for s in ("re", "im") for l in (0, 1) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    sd = {}
    for n in names:
        l = int(n[-1])
        shape = (4 * H, I if l == 0 else H) if "weight_ih" in n else ((4 * H, H) if "weight_hh" in n else (4 * H,))
        sd[n] = O.synth_tensor(n, shape, 5) * (3.0 if "weight" in n else 1.0)
    want = O.complex_lstm(x.double(), {k: v.double() for k, v in sd.items()}, "", 2).float()
    xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).cuda())
    get = lambda n: sd[n].cuda()
    p0, p1 = ops.pack_lstm(get, H, I, 0, "cuda"), ops.pack_lstm(get, H, H, 1, "cuda")
    try:
        ops.set_precision("bf16x3")
        got16 = ops.clstm(xp, p0, p1, H).channel_slice(0, H).cpu().permute(1, 0, 2, 3)
    finally:
        ops.set_precision("fp32")
    got32 = ops.clstm(xp, p0, p1, H).channel_slice(0, H).cpu().permute(1, 0, 2, 3)
    e16, e32 = relerr(got16, want), relerr(got32, want)
    assert e32 < TOL and e16 < 1e-4, (e16, e32)


@pytest.mark.parametrize("cin,skip_c,F,T,B", [(16, 16, 17, 70, 2), (32, 32, 129, 40, 3), (8, 0, 9, 130, 2), (24, 8, 33, 33, 1)])
def test_ctconv_c1_bf16x3(ops, cin, skip_c, F, T, B):
    """Cout = 1 transposed conv with the frequency taps re-associated into the MFMA M dimension (last decoder block)."""
    g = torch.Generator().manual_seed(21)
    dev = "cuda"
    cin_tot = cin + skip_c
    x = torch.randn(B, cin, F, T, 2, generator=g)
    wr, wi = torch.randn(cin_tot, 1, 5, 2, generator=g) * 0.1, torch.randn(cin_tot, 1, 5, 2, generator=g) * 0.1
    br, bi = torch.randn(1, generator=g), torch.randn(1, generator=g)
    xin, sk = x, None
    if skip_c:
        sk = torch.randn(B, skip_c, F, T, 2, generator=g)
        xin = torch.cat([x, sk], dim=1)
    want = O.complex_conv_transpose2d(xin.double(), wr.double(), br.double(), wi.double(), bi.double(), (2, 1), (2, 0), True).float()
    mom = torch.tensor([[0.05], [-0.02], [0.8], [0.1], [1.2]])
    gam = [torch.tensor([1.1]), torch.tensor([0.3]), torch.tensor([0.9])]
    bet = [torch.tensor([0.05]), torch.tensor([-0.1])]
    want = O.prelu(O.cbn_whiten_affine(want, mom[0], mom[1], mom[2], mom[3], mom[4], gam[0], gam[1], gam[2], bet[0], bet[1]),
                   torch.tensor(0.25))
    fold = ops.cbn_fold(mom.to(dev), *[t.to(dev) for t in gam], *[t.to(dev) for t in bet])
    xp = ops.Planar.from_tensor5(x.to(dev), T + 2)
    skp = ops.Planar.from_tensor5(sk.to(dev), T + 2) if sk is not None else None
    _, bias = ops.pack_cconv(wr.to(dev), wi.to(dev), br.to(dev), bi.to(dev), fold, transposed=True)
    wc1 = ops.pack_ctconv_c1(wr.to(dev), wi.to(dev), fold)
    y = ops.ctconv_c1(xp, wc1, bias, slope=torch.tensor([0.25], device=dev), skip=skp)
    got = y.tensor5().cpu()
    assert got.shape == want.shape
    assert relerr(got, want) < 2e-4
    assert float(y.planes()[..., 0].abs().max()) == 0.0 and float(y.planes()[..., y.T + 1:].abs().max()) == 0.0
    # split-image sources: same operand split, identical result
    yi = ops.ctconv_c1(ops.to_image(xp), wc1, bias, slope=torch.tensor([0.25], device=dev),
                       skip=ops.to_image(skp) if skp is not None else None)
    assert torch.equal(yi.tensor5(), y.tensor5())
    assert float(yi.planes()[..., 0].abs().max()) == 0.0


@pytest.mark.parametrize("L,B", [(3201, 3), (64000, 2)])
def test_stft_istft_dense_bf16x3(ops, L, B):
    """bf16x3 point-wise contractions (K-major split image + idv_pw_bf16x3): windowed DFT, inverse DFT and
    ComplexDense against the oracle; frame indexing exact, guard columns zero, STFT -> ISTFT identity."""
    g = torch.Generator().manual_seed(L)
    x = torch.randn(B, L, generator=g) * 0.1
    keep = ops.PRECISION
    try:
        ops.set_precision("bf16x3")
        T = 1 + L // HOP
        plan = ops.DftPlan(NFFT, WIN, HOP, T, "cuda")
        X = ops.stft(x.cuda(), plan)
        assert X.T == T and X.F == NFFT // 2 + 1
        assert relerr(X.tensor4().cpu(), O.stft(x, NFFT, HOP, WIN)) < 2e-5
        assert float(X.planes()[..., 0].abs().max()) == 0.0
        y = ops.istft(X, plan).cpu()
        assert y.shape[1] == HOP * (L // HOP)
        assert relerr(y, x[:, :y.shape[1]]) < 2e-5
        # dense 128 -> 1280 (the DCCRN bottleneck shape)
        K, M, Td = 128, 1280, 37
        xd = torch.randn(B, K, 1, Td, 2, generator=g)
        wr, br = torch.randn(M, K, generator=g) * 0.1, torch.randn(M, generator=g)
        wi, bi = torch.randn(M, K, generator=g) * 0.1, torch.randn(M, generator=g)
        xp = ops.Planar.from_tensor5(xd.cuda())
        pr = ops.pack_pw(wr.cuda(), br.cuda()) + (ops.pack_pw_bf16(wr.cuda()),)
        pi = ops.pack_pw(wi.cuda(), bi.cuda()) + (ops.pack_pw_bf16(wi.cuda()),)
        out = ops.cdense(xp, pr, pi, M, 256, 5)
        got = out.tensor5().cpu()                                          # [B, 256, 5, T, 2]
        xr, xi = xd[:, :, 0, :, 0].permute(0, 2, 1).double(), xd[:, :, 0, :, 1].permute(0, 2, 1).double()   # [B, T, K]
        want_r = (xr @ wr.double().T + br.double()).reshape(B, Td, 256, 5).permute(0, 2, 3, 1)
        want_i = (xi @ wi.double().T + bi.double()).reshape(B, Td, 256, 5).permute(0, 2, 3, 1)
        assert relerr(got[..., 0], want_r.float()) < 2e-5 and relerr(got[..., 1], want_i.float()) < 2e-5
        assert float(out.planes()[..., 0].abs().max()) == 0.0
    finally:
        ops.set_precision(keep)


@pytest.mark.parametrize("precision,B,T,cin,cout,F", [("fp32", 3, 50, 64, 32, 17), ("bf16x3", 3, 50, 64, 32, 17),
                                                      ("bf16x3", 16, 641, 256, 128, 9)])
def test_conv_adjoint_identity(ops, precision, B, T, cin, cout, F):
    """Size-independent property (no oracle): the causal transposed conv y = A x and the non-causal conv with the
    conjugate-transposed weights are adjoint, <A x, y> = <x, A* y>, and A is linear -- checked through the C ABI at
    a full-size decoder shape as well.  (A* is also the data gradient the backward row will use.)"""
    g = torch.Generator().manual_seed(5)
    dev = "cuda"
    wr, wi = torch.randn(cin, cout, 5, 2, generator=g) * 0.05, torch.randn(cin, cout, 5, 2, generator=g) * 0.05
    zb_o, zb_i = torch.zeros(cout), torch.zeros(cin)
    x = torch.randn(B, cin, F, T, 2, generator=g)
    x[:, :, :, T - 1] = 0.0                       # the non-causal conv yields frames 0 .. T-2
    x2 = torch.randn(B, cin, F, T, 2, generator=g)
    x2[:, :, :, T - 1] = 0.0
    y = torch.randn(B, cout, 2 * F - 1, T, 2, generator=g)
    Tp = T + 1
    xp, x2p, yp = (ops.Planar.from_tensor5(t.to(dev), Tp) for t in (x, x2, y))
    wfA, bA = ops.pack_cconv(wr.to(dev), wi.to(dev), zb_o.to(dev), zb_o.to(dev), None, transposed=True)
    # A*: conv weights [Cout' = cin][Cin' = cout] = the transposed-conv tensor itself, imaginary part negated
    wfT, bT = ops.pack_cconv(wr.to(dev), (-wi).to(dev), zb_i.to(dev), zb_i.to(dev), None, transposed=False)
    keep = ops.PRECISION
    try:
        ops.set_precision(precision)
        kwA = dict(transposed=True, causal=True)
        kwT = dict(transposed=False, causal=False)
        if precision == "bf16x3":
            kwA["wfrag_bf16"] = ops.pack_cconv_bf16(wr.to(dev), wi.to(dev), None, transposed=True)
            kwT["wfrag_bf16"] = ops.pack_cconv_bf16(wr.to(dev), (-wi).to(dev), None, transposed=False)
        Ax = ops.cconv2d(xp, wfA, bA, cout, **kwA)
        Ax2 = ops.cconv2d(x2p, wfA, bA, cout, **kwA)
        Aty = ops.cconv2d(yp, wfT, bT, cin, **kwT)
        assert Aty.T == T - 1 and Ax.T == T
        lhs = float((Ax.tensor5().double() * y.to(dev).double()).sum())
        rhs = float((Aty.tensor5().double() * x[:, :, :, :T - 1].to(dev).double()).sum())
        scale = float(Ax.tensor5().double().norm() * y.double().norm())
        tol = 1e-6 if precision == "fp32" else 2e-5
        assert abs(lhs - rhs) < tol * scale, (lhs, rhs, scale)
        # the other pair: causal conv C (encoder block) and the transposed conv reading (x[t+1], x[t]) with the
        # conjugate-transposed weights.  conv weight [cout2][cin2] = the tensor above read as [cin][cout]
        cw_r, cw_i = wr, wi                                               # conv: Cout2 = cin, Cin2 = cout
        u = torch.randn(B, cout, 2 * F - 1, T, 2, generator=g)            # conv input  [B, Cin2, 2F-1, T]
        v = torch.randn(B, cin, F, T, 2, generator=g)                     # conv output [B, Cout2, F, T]
        up, vp = ops.Planar.from_tensor5(u.to(dev), Tp), ops.Planar.from_tensor5(v.to(dev), Tp)
        wfC, bC = ops.pack_cconv(cw_r.to(dev), cw_i.to(dev), zb_i.to(dev), zb_i.to(dev), None, transposed=False)
        wfCt, bCt = ops.pack_cconv(cw_r.to(dev), (-cw_i).to(dev), zb_o.to(dev), zb_o.to(dev), None, transposed=True)
        kwC, kwCt = dict(transposed=False, causal=True), dict(transposed=True, causal=True, adjoint_time=True)
        if precision == "bf16x3":
            kwC["wfrag_bf16"] = ops.pack_cconv_bf16(cw_r.to(dev), cw_i.to(dev), None, transposed=False)
            kwCt["wfrag_bf16"] = ops.pack_cconv_bf16(cw_r.to(dev), (-cw_i).to(dev), None, transposed=True)
        Cu = ops.cconv2d(up, wfC, bC, cin, **kwC)
        Ctv = ops.cconv2d(vp, wfCt, bCt, cout, **kwCt)
        assert Cu.T == T and Ctv.T == T and Ctv.F == 2 * F - 1
        lhs2 = float((Cu.tensor5().double() * v.to(dev).double()).sum())
        rhs2 = float((Ctv.tensor5().double() * u.to(dev).double()).sum())
        scale2 = float(Cu.tensor5().double().norm() * v.double().norm())
        assert abs(lhs2 - rhs2) < tol * scale2, (lhs2, rhs2, scale2)
        # linearity: A(x + 2 x2) = A x + 2 A x2
        xs = ops.Planar.from_tensor5((x + 2 * x2).to(dev), Tp)
        Axs = ops.cconv2d(xs, wfA, bA, cout, **kwA)
        want = Ax.tensor5() + 2 * Ax2.tensor5()
        assert relerr(Axs.tensor5(), want) < (2e-6 if precision == "fp32" else 2e-5)
    finally:
        ops.set_precision(keep)
