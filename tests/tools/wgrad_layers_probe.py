"""Per-layer timing of the complex weight gradient at the DCCRN-CL train-step shapes (B utterances of 4 s): which layers sit
below the others.  python tests/tools/wgrad_layers_probe.py [B]   (GPU box; prints ms and executed / algorithmic TFLOP/s)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops
L = amd._lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 641
ENC = [(1, 32), (32, 64), (64, 128), (128, 128), (128, 256), (256, 256)]          # model/causal_netconfig.py
DEC = [(512, 256), (512, 128), (256, 128), (256, 64), (128, 32), (64, 1)]         # input = torch.cat(previous, skip)
FE = [257, 129, 65, 33, 17, 9, 5]
dev = "cuda"
rows = []
for tr, layers in ((False, ENC), (True, DEC)):
    for k, (cin, cout) in enumerate(layers):
        fin = FE[6 - k] if tr else FE[k]
        fout = FE[5 - k] if tr else FE[k + 1]
        x = ops.Planar.empty(cin, fin, B, T, T + 1, dev, zero=True)
        dy = ops.Planar.empty(cout, fout, B, T, T + 1, dev, zero=True)
        x.tensor5().normal_()
        dy.tensor5().normal_()
        shp = (cin, cout, 5, 2) if tr else (cout, cin, 5, 2)
        dw_re = torch.empty(shp, device=dev)
        dw_im = torch.empty(shp, device=dev)
        cs, cl = (cin, cout) if tr else (cout, cin)
        gauss = bool(L.lib().idv_cconv_wgrad_gauss_supported(L.i(cs), L.i(cl)))
        for _ in range(2):
            ops.cconv_wgrad(x, 0, dy, cout, cin, tr, True, dw_re, dw_im)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            ops.cconv_wgrad(x, 0, dy, cout, cin, tr, True, dw_re, dw_im)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        macs = 4 * cin * cout * 10 * B * T * (fin if tr else fout)
        rows.append((("dec" if tr else "enc") + str(k), cin, cout, fin, fout, gauss, ms, 2 * macs / ms / 1e9))
        del x, dy
tot = 0.0
for name, cin, cout, fin, fout, gauss, ms, tf in rows:
    tot += ms
    print(f"{name}  {cin:4d}->{cout:4d}  F {fin:3d}->{fout:3d}  {'gauss' if gauss else 'four '}  {ms:7.3f} ms  {tf:6.1f} TF algorithmic"
          f"  {tf * (0.75 if gauss else 1.0):6.1f} executed", flush=True)
print(f"total {tot:.2f} ms")
