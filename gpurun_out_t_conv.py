import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import idccrn_vae_amd as A
from idccrn_vae_amd import ops
from idccrn_vae_amd.ops import Planar
from oracle import idccrn_oracle as O
dev = 'cuda'
torch.manual_seed(0)
def rel(a, b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
for (causal, transposed, cin, cout, F, T, B) in [(True, False, 4, 8, 17, 9, 2), (True, False, 1, 32, 33, 40, 3), (False, False, 4, 8, 17, 9, 2),
                                                  (True, True, 6, 4, 9, 9, 2), (False, True, 6, 4, 9, 9, 2), (True, True, 8, 1, 17, 50, 3),
                                                  (True, False, 32, 64, 65, 70, 2), (True, True, 64, 32, 9, 70, 2)]:
    x = torch.randn(B, cin, F, T, 2)
    if transposed:
        wr, wi = torch.randn(cin, cout, 5, 2)*0.2, torch.randn(cin, cout, 5, 2)*0.2
    else:
        wr, wi = torch.randn(cout, cin, 5, 2)*0.2, torch.randn(cout, cin, 5, 2)*0.2
    br, bi = torch.randn(cout), torch.randn(cout)
    pad = (2, 1) if (causal and not transposed) else (2, 0)
    if transposed:
        want = O.complex_conv_transpose2d(x, wr, br, wi, bi, (2, 1), (2, 0), causal)
    else:
        want = O.complex_conv2d(x, wr, br, wi, bi, (2, 1), pad, causal)
    Tp = T + 2
    xp = Planar.from_tensor5(x.to(dev), Tp)
    wfrag, bias = ops.pack_cconv(wr.to(dev), wi.to(dev), br.to(dev), bi.to(dev), None, transposed=transposed)
    y = ops.cconv2d(xp, wfrag, bias, cout, transposed=transposed, causal=causal)
    torch.cuda.synchronize()
    got = y.tensor5().cpu()
    print(f"causal={causal} T={transposed} cin={cin} cout={cout} F={F} T={T}: shape {tuple(got.shape)} vs {tuple(want.shape)} rel={rel(got, want):.2e}",
          "guard0=", float(y.planes()[..., 0].abs().max()), "guardT=", float(y.planes()[..., y.T+1:].abs().max()))
