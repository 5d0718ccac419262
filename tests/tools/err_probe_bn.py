"""usage (GPU box): python tests/tools/err_probe_bn.py -- train-mode CBN + PReLU backward in isolation on the REAL pre-BN
activations of every block of the full-width DCCRN-CL (random upstream gradient): HIP vs float64 autograd, float32 vs float64,
with the conditioning of each block's worst channel."""
import importlib
import sys
import torch

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_backward as T
from oracle import idccrn_oracle as O
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops
AG = importlib.import_module("i-dccrn-vae_amd.autograd")
pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
NFFT, HOP, WIN, SKIP = 512, 100, 400, [0, 1, 2, 3, 4, 5]
np_ = O.net_params(True, 32)
m = T.load_synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None), 77)
g = torch.Generator().manual_seed(21)
x = torch.randn(2, 16000, generator=g) * 0.1
ys = []
orig = ops.cbn_apply_to
def spy(y, fold, slope):
    ys.append((y, fold))
    return orig(y, fold, slope)
ops.cbn_apply_to = spy
with torch.enable_grad():
    m(x.cuda(), train=True)
ops.cbn_apply_to = orig
blocks = list(m.std_DCCRN.encoders) + list(m.std_DCCRN.decoders)
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / (b.detach().cpu().double().norm() + 1e-30))
for idx, ((y, fold), blk) in enumerate(zip(ys, blocks)):
    bn = blk.bn
    y5 = y.tensor5().detach().cpu()
    R = torch.randn(y5.shape, generator=g)
    # HIP: stats -> finalize -> bwd
    stats = ops.cbn_stats(y)
    moments = torch.empty(5, bn.C, device="cuda"); fold2 = torch.empty(bn.C, 6, device="cuda")
    from importlib import import_module
    L = amd._lib
    L.call("idv_cbn_finalize", L.p(stats), L.d(float(y.B) * y.F * y.T), L.p(bn.gamma_rr), L.p(bn.gamma_ri), L.p(bn.gamma_ii), L.p(bn.beta_r),
           L.p(bn.beta_i), L.i(bn.C), L.i(1), L.f(0.9), L.p(None), L.p(None), L.p(None), L.p(None), L.p(None), L.p(moments), L.p(fold2),
           L.stream_ptr())
    dz = ops.Planar.from_tensor5(R.cuda(), y.Tp)
    slope = blk.prelu.weight.detach()
    dy = ops.cbn_bwd(dz, y, fold2, moments, (bn.gamma_rr.detach(), bn.gamma_ri.detach(), bn.gamma_ii.detach()), slope, float(y.B) * y.F * y.T)[0]
    out = {}
    for dt in (torch.float64, torch.float32):
        yy = y5.to(dt).clone().requires_grad_(True)
        st = O.cbn_batch_stats(yy)
        u = O.cbn_whiten_affine(yy, *st, *[t.detach().cpu().to(dt) for t in (bn.gamma_rr, bn.gamma_ri, bn.gamma_ii, bn.beta_r, bn.beta_i)])
        z = O.prelu(u, slope.cpu().to(dt))
        (z * R.to(dt)).sum().backward()
        out[dt] = (yy.grad, st)
    st = out[torch.float64][1]
    Vrr, Vri, Vii = (t.reshape(-1) for t in st[2:])
    cond = (Vrr * Vii / (Vrr * Vii - Vri * Vri + 1e-5)).max()
    ratio = (torch.stack([st[0].reshape(-1).abs() / Vrr.sqrt(), st[1].reshape(-1).abs() / Vii.sqrt()])).max()
    print(f"block {idx} C={bn.C}: dy HIP {rel(dy.tensor5(), out[torch.float64][0]):.1e} f32 {rel(out[torch.float32][0], out[torch.float64][0]):.1e} | "
          f"max VrrVii/det {float(cond):.1e}  max |mean|/std {float(ratio):.1e}  min V {float(torch.minimum(Vrr, Vii).min()):.1e}")
