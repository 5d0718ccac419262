"""Diagnostic: is the three-product conv kernel's result for an utterance independent of the batch it sits in?
(B = 4 run vs the two B = 2 halves, bitwise; forward, train-mode statistics, adjoint.)  python tests/tools/gauss_batch_probe.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ops = importlib.import_module("i-dccrn-vae_amd").ops
dev = "cuda"
g = torch.Generator().manual_seed(3)
T = 17
shapes = [(False, 4, 8, 129, 0), (False, 8, 16, 65, 0), (False, 16, 16, 33, 0), (False, 16, 32, 17, 0), (False, 32, 32, 9, 0),
          (True, 32, 16, 5, 32), (True, 16, 16, 9, 16), (True, 16, 8, 17, 16), (True, 8, 4, 33, 8), (True, 4, 4, 65, 4)]
for transposed, cin, cout, F, skc in shapes:
    x = torch.randn(4, cin, F, T, 2, generator=g)
    sk = torch.randn(4, skc, F, T, 2, generator=g) if skc else None
    ct = cin + skc
    shape = (ct, cout, 5, 2) if transposed else (cout, ct, 5, 2)
    wr, wi = (torch.randn(shape, generator=g) * 0.2).to(dev), (torch.randn(shape, generator=g) * 0.2).to(dev)
    br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
    g3 = ops.pack_cconv_gauss(wr, wi, br, bi, None, transposed=transposed)
    wf, bs = ops.pack_cconv(wr, wi, br, bi, None, transposed=transposed)

    def run(xs, sks, gauss, stats):
        xp = ops.Planar.from_tensor5(xs.to(dev), T + 1)
        sp = ops.Planar.from_tensor5(sks.to(dev), T + 1) if sks is not None else None
        st = torch.zeros(cout, 5, dtype=torch.float64, device=dev) if stats else None
        if gauss:
            y = ops.cconv2d(xp, None, None, cout, transposed=transposed, skip=sp, stats=st, gauss=g3)
        else:
            y = ops.cconv2d(xp, wf, bs, cout, transposed=transposed, skip=sp, stats=st)
        return y.tensor5().clone(), st
    for gauss in (True, False):
        for stats in (False, True):
            full, sf = run(x, sk, gauss, stats)
            h0, s0 = run(x[:2], sk[:2] if sk is not None else None, gauss, stats)
            h1, s1 = run(x[2:], sk[2:] if sk is not None else None, gauss, stats)
            eq = torch.equal(full[:2], h0) and torch.equal(full[2:], h1)
            d = max(float((full[:2] - h0).abs().max()), float((full[2:] - h1).abs().max()))
            serr = float(((s0 + s1) - sf).abs().max() / sf.abs().max()) if stats else 0.0
            print(f"tr={transposed} cin={cin}+{skc} cout={cout} F={F} gauss={gauss} stats={stats}: bit-equal={eq} maxdiff={d:.2e} stats rel={serr:.1e}")
    # adjoint
    Fo = 2 * F - 1 if transposed else (F - 1) // 2 + 1
    dy = torch.randn(4, cout, Fo, T, 2, generator=g)
    ga = ops.pack_cconv_gauss(wr[:cin] if transposed else wr, wi[:cin] if transposed else wi, None, None, None,
                              adjoint_of=(cin, cout, cout, not transposed)) if not transposed or skc == 0 else None
    if ga is not None:
        def radj(d_):
            return ops.cconv_dgrad(ops.Planar.from_tensor5(d_.to(dev), T + 1), None, None, cin, transposed, True, gauss=ga).tensor5().clone()
        full = radj(dy)
        eq = torch.equal(full[:2], radj(dy[:2])) and torch.equal(full[2:], radj(dy[2:]))
        print(f"   adjoint bit-equal={eq}")
