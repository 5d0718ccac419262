"""Per-layer timing of the DCCRN-CL decoder's transposed convs (dec0 .. dec4 at B utterances of 4 s) on the Winograd kernel
(csrc/cgemm_wino.hip) beside cgemm_gauss_kernel.  IDV_WINO_CFG=<WM WN CIK digits> selects an experimental workgroup shape
(read once per process).   python tests/tools/wino_layers_probe.py [B]      (GPU box)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = 641
ENC = [(32, 64), (64, 128), (128, 128), (128, 256), (256, 256)]              # enc1 .. enc5
DEC = [(512, 256), (512, 128), (256, 128), (256, 64), (128, 32)]
FE = [257, 129, 65, 33, 17, 9, 5]
dev = "cuda"
g = torch.Generator().manual_seed(0)
slope = torch.tensor([0.25], device=dev)
tot = [0.0, 0.0]
line = []
tag = os.environ.get("IDV_WINO_CFG", "default") + "/" + os.environ.get("IDV_WINO_CCFG", "default")
for tr, layers in ((False, ENC), (True, DEC)):
    for k, (cin, cout) in enumerate(layers):
        fin = FE[6 - k] if tr else FE[k + 1]
        x = ops.Planar.empty(cin, fin, B, T, T + 1, dev, zero=True)
        x.tensor5().normal_()
        shape = (cin, cout, 5, 2) if tr else (cout, cin, 5, 2)
        wr, wi = torch.randn(shape, generator=g).to(dev) * 0.05, torch.randn(shape, generator=g).to(dev) * 0.05
        br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
        ops.WINO = True
        pk = ops.pack_cconv_gauss(wr, wi, br, bi, None, transposed=tr)
        res = []
        for wino in (False, True):
            ops.WINO = wino
            for _ in range(2):
                y = ops.cconv2d(x, None, None, cout, transposed=tr, slope=slope, gauss=pk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 4
            e0.record()
            for _ in range(n):
                y = ops.cconv2d(x, None, None, cout, transposed=tr, slope=slope, gauss=pk)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / n)
        tot[0] += res[0]
        tot[1] += res[1]
        line.append(f"{'dec' if tr else 'enc'}{k if tr else k + 1} {res[0]:6.2f} -> {res[1]:6.2f}")
        del x, y
print(f"[wino cfg {tag}] " + " | ".join(line) + f" | total {tot[0]:.2f} -> {tot[1]:.2f} ms", flush=True)
