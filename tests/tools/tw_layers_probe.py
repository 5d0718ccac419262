"""Per-layer timing of the DCCRN-CL decoder's transposed convs (dec0 .. dec3 at B utterances of 4 s) on the time-Winograd kernels
(csrc/cgemm_tw.hip) beside cgemm_wino.     python tests/tools/tw_layers_probe.py [B]      (GPU box)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = 641
DEC = [(256, 256, 256), (256, 256, 128), (128, 128, 128), (128, 128, 64)]     # (from below, skip, out)
FE = [5, 9, 17, 33]
dev = "cuda"
g = torch.Generator().manual_seed(0)
slope = torch.tensor([0.25], device=dev)
tot = [0.0, 0.0]
line = []
for k, (c0, c1, cout) in enumerate(DEC):
    fin = FE[k]
    x = ops.Planar.empty(c0, fin, B, T, T + 1, dev, zero=True)
    x.tensor5().normal_()
    sk = ops.Planar.empty(c1, fin, B, T, T + 1, dev, zero=True)
    sk.tensor5().normal_()
    shape = (c0 + c1, cout, 5, 2)
    wr, wi = torch.randn(shape, generator=g).to(dev) * 0.05, torch.randn(shape, generator=g).to(dev) * 0.05
    br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
    ops.WINO = ops.TW = True
    pk = ops.pack_cconv_gauss(wr, wi, br, bi, None, transposed=True)
    res, outs = [], []
    for tw in (False, True):
        ops.TW = tw
        for _ in range(2):
            y = ops.cconv2d(x, None, None, cout, transposed=True, slope=slope, skip=sk, gauss=pk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = ops.cconv2d(x, None, None, cout, transposed=True, slope=slope, skip=sk, gauss=pk)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
        outs.append(y.tensor5().clone())
    err = float((outs[1] - outs[0]).norm() / outs[0].norm())
    tot[0] += res[0]
    tot[1] += res[1]
    line.append(f"dec{k} {res[0]:6.2f} -> {res[1]:6.2f} ({err:.1e})")
print("[wino -> time-Winograd] " + " | ".join(line) + f" | total {tot[0]:.2f} -> {tot[1]:.2f} ms")
# encoder convs enc1 .. enc5 on the conv form (csrc/cgemm_tw2.hip)
ENC = [(32, 64), (64, 128), (128, 128), (128, 256), (256, 256)]
FI = [129, 65, 33, 17, 9]
tot, line = [0.0, 0.0], []
for k, (cin, cout) in enumerate(ENC):
    x = ops.Planar.empty(cin, FI[k], B, T, T + 1, dev, zero=True)
    x.tensor5().normal_()
    wr, wi = torch.randn((cout, cin, 5, 2), generator=g).to(dev) * 0.05, torch.randn((cout, cin, 5, 2), generator=g).to(dev) * 0.05
    br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
    ops.WINO = ops.TW = ops.TW_CONV = True
    pk = ops.pack_cconv_gauss(wr, wi, br, bi, None, transposed=False)
    res, outs = [], []
    for tw in (False, True):
        ops.TW_CONV = tw
        for _ in range(2):
            y = ops.cconv2d(x, None, None, cout, transposed=False, slope=slope, gauss=pk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = ops.cconv2d(x, None, None, cout, transposed=False, slope=slope, gauss=pk)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
        outs.append(y.tensor5().clone())
    err = float((outs[1] - outs[0]).norm() / outs[0].norm())
    tot[0] += res[0]
    tot[1] += res[1]
    line.append(f"enc{k + 1} {res[0]:6.2f} -> {res[1]:6.2f} ({err:.1e})")
print("[wino conv -> time-Winograd conv] " + " | ".join(line) + f" | total {tot[0]:.2f} -> {tot[1]:.2f} ms")
