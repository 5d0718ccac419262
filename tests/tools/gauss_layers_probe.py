"""Per-layer timing of the three-product complex conv / transposed conv at the DCCRN-CL evaluation shapes (B utterances of 4 s).
python tests/tools/gauss_layers_probe.py [B]      (GPU box; ms and algorithmic / executed TFLOP/s per layer)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = 641
ENC = [(32, 64), (64, 128), (128, 128), (128, 256), (256, 256)]              # enc1 .. enc5 (enc0: one input channel, not this kernel)
DEC = [(512, 256), (512, 128), (256, 128), (256, 64), (128, 32)]             # dec0 .. dec4 (dec5: one output channel)
FE = [257, 129, 65, 33, 17, 9, 5]
dev = "cuda"
g = torch.Generator().manual_seed(0)
slope = torch.tensor([0.25], device=dev)
tot = 0.0
for tr, layers in ((False, ENC), (True, DEC)):
    for k, (cin, cout) in enumerate(layers):
        fin = FE[6 - k] if tr else FE[k + 1]
        fout = FE[5 - k] if tr else FE[k + 2]
        x = ops.Planar.empty(cin, fin, B, T, T + 1, dev, zero=True)
        x.tensor5().normal_()
        shape = (cin, cout, 5, 2) if tr else (cout, cin, 5, 2)
        wr, wi = torch.randn(shape, generator=g).to(dev) * 0.05, torch.randn(shape, generator=g).to(dev) * 0.05
        br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
        mom = torch.stack([torch.zeros(cout), torch.zeros(cout), torch.ones(cout), torch.zeros(cout), torch.ones(cout)]).to(dev)
        one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        fold = ops.cbn_fold(mom, one, zero, one, zero, zero)
        pk = ops.pack_cconv_gauss(wr, wi, br, bi, fold, transposed=tr)
        for _ in range(2):
            y = ops.cconv2d(x, None, None, cout, transposed=tr, slope=slope, gauss=pk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            y = ops.cconv2d(x, None, None, cout, transposed=tr, slope=slope, gauss=pk)
        e1.record()
        torch.cuda.synchronize()
        assert y.F == fout, (y.F, fout)
        ms = e0.elapsed_time(e1) / n
        tot += ms
        macs = 4 * cin * cout * 10 * B * T * (fin if tr else fout)
        tf = 2 * macs / ms / 1e9
        print(f"{'dec' if tr else 'enc'}{k if tr else k + 1}  {cin:4d}->{cout:4d}  F {fin:3d}->{fout:3d}  {ms:7.3f} ms  {tf:6.1f} TF algorithmic"
              f"  {0.75 * tf:6.1f} executed",
              flush=True)
        del x, y
print(f"total {tot:.2f} ms")
