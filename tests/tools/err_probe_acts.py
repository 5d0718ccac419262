"""usage (GPU box): python tests/tools/err_probe_acts.py -- error of the ACTIVATION gradients (d loss / d block output) of the
full-width DCCRN-CL train step, block by block in backward order: HIP vs float64 oracle, float32 oracle vs float64."""
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
nl = importlib.import_module("i-dccrn-vae_amd.model.nsvae_loss")
NFFT, HOP, WIN, SKIP = 512, 100, 400, [0, 1, 2, 3, 4, 5]
np_ = O.net_params(True, 32)
m = T.load_synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None), 77)
g = torch.Generator().manual_seed(21)
x = torch.randn(2, 16000, generator=g) * 0.1
c = x + torch.randn(2, 16000, generator=g) * 0.05
w = [0.2, 0.1, 1.0]

hip = []          # (name, Planar, grad buf) in forward order
orig_cb = AG.conv_block
def cb(conv, bn, prelu_weight, x_, skip, zero_skip):
    z = orig_cb(conv, bn, prelu_weight, x_, skip, zero_skip)
    rec = [f"{'dec' if conv._transposed else 'enc'}{sum(1 for h in hip if h[0].startswith('dec' if conv._transposed else 'enc'))}", z, None]
    z.buf.register_hook(lambda gr, rec=rec: rec.__setitem__(2, gr.detach().clone()))
    hip.append(rec)
    return z
AG.conv_block = cb
pm.AG.conv_block = cb
with torch.enable_grad():
    est, pred = m(x.cuda(), train=True)
    nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(c.cuda()), c.cuda(), est)[0].backward()


def run(dt):
    acts = []
    oe, od = O.encoder_block, O.decoder_block
    def enc(*a, **k):
        y = oe(*a, **k); y.retain_grad(); acts.append((f"enc{sum(1 for n, _ in acts if n.startswith('enc'))}", y)); return y
    def dec(*a, **k):
        y = od(*a, **k); y.retain_grad(); acts.append((f"dec{sum(1 for n, _ in acts if n.startswith('dec'))}", y)); return y
    O.encoder_block, O.decoder_block = enc, dec
    try:
        sd = {k: v.detach().cpu().to(dt).clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and k[-3:] not in ("Vrr", "Vri", "Vii"))
              for k, v in m.state_dict().items()}
        e, p, _ = O.dccrn_forward(x.to(dt), sd, np_, True, NFFT, HOP, WIN, SKIP, "mask", True, O.BNState())
        O.multiple_recon_loss(p, O.stft(c.to(dt), NFFT, HOP, WIN), c.to(dt), e, w)[0].backward()
    finally:
        O.encoder_block, O.decoder_block = oe, od
    return dict(acts)


rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / (b.detach().cpu().double().norm() + 1e-30))
a64, a32 = run(torch.float64), run(torch.float32)
for name, z, gr in reversed(hip):
    gz = ops.rewrap(gr, z).tensor5()
    print(f"{name}: forward HIP {rel(z.tensor5(), a64[name]):.1e} f32 {rel(a32[name], a64[name]):.1e} | grad HIP {rel(gz, a64[name].grad):.1e} f32 {rel(a32[name].grad, a64[name].grad):.1e}")
