"""Outputs (forward with moment sums, data gradient) of the three-product conv kernels at the DCCRN-CL training shapes, saved per
configuration:  IDV_GAUSS_CCFG=0 IDV_GAUSS_OCC2_MAXC=0 IDV_GAUSS_TWM=2 IDV_STATS_REP=1 python tests/tools/conv_cfg_compare.py b ;
python tests/tools/conv_cfg_compare.py a ; python tests/tools/conv_cfg_compare.py compare   (GPU box; writes gpurun_out/cmp_*.pt).
The tile configurations give BIT-identical y and dx; only the moment sums differ (1e-9: grouping of the fp32 partials)."""
import importlib, os, sys, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
tag = sys.argv[1]
if tag == "compare":
    a, b = torch.load("gpurun_out/cmp_a.pt"), torch.load("gpurun_out/cmp_b.pt")
    for k in a:
        d = (a[k].double() - b[k].double()).norm() / b[k].double().norm()
        print(f"{k:40s} rel {float(d):.2e}  max|a| {float(a[k].abs().max()):.3e}")
    sys.exit(0)
amd = importlib.import_module("i-dccrn-vae_amd"); ops = amd.ops
B, T, dev = 2, 400, "cuda"
g = torch.Generator().manual_seed(0)
ENC = [(32, 64), (64, 128), (128, 128), (128, 256), (256, 256)]
DEC = [(512, 256), (512, 128), (256, 128), (256, 64), (128, 32)]
FE = [257, 129, 65, 33, 17, 9, 5]
out = {}
for tr, layers in ((False, ENC), (True, DEC)):
    for k, (cin, cout) in enumerate(layers):
        fin = FE[6 - k] if tr else FE[k + 1]
        x = ops.Planar.from_tensor5(torch.randn(B, cin, fin, T, 2, generator=g).to(dev), T + 1)
        shape = (cin, cout, 5, 2) if tr else (cout, cin, 5, 2)
        wr, wi = (torch.randn(shape, generator=g) * 0.05).to(dev), (torch.randn(shape, generator=g) * 0.05).to(dev)
        br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
        pk = ops.pack_cconv_gauss(wr, wi, br, bi, None, transposed=tr)
        st = torch.zeros(cout, 5, dtype=torch.float64, device=dev)
        y = ops.cconv2d(x, None, None, cout, transposed=tr, gauss=pk, stats=st)
        name = f"{'dec' if tr else 'enc'}{k if tr else k + 1}"
        out[name + ".y"] = y.tensor5().cpu().clone()
        out[name + ".stats"] = st.cpu().clone()
        dy = ops.Planar.from_tensor5(torch.randn(B, cout, y.F, T, 2, generator=g).to(dev), T + 1)
        ga = ops.pack_cconv_gauss(wr, wi, None, None, None, adjoint_of=(cin, cout, cout, not tr))
        dx = ops.cconv_dgrad(dy, None, None, cin, tr, True, gauss=ga)
        out[name + ".dx"] = dx.tensor5().cpu().clone()
torch.save(out, f"gpurun_out/cmp_{tag}.pt")
print("saved", tag, len(out))
