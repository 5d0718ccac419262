"""usage (GPU box): python tests/tools/err_probe_block.py -- every block's backward in isolation on the REAL inputs and the REAL
upstream gradient of the full-width train step: data / skip gradient of the HIP block vs the float64 (and float32) oracle block."""
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
recs = []
orig_cb = AG.conv_block
def cb(conv, bn, prelu_weight, x_, skip, zero_skip):
    z = orig_cb(conv, bn, prelu_weight, x_, skip, zero_skip)
    rec = dict(conv=conv, bn=bn, x=x_, skip=skip, z=z, dz=None, dx=None, dskip=None)
    z.buf.register_hook(lambda gr, rec=rec: rec.__setitem__("dz", gr.detach().clone()))
    if x_.buf.requires_grad:
        x_.buf.register_hook(lambda gr, rec=rec: rec.__setitem__("dx_total", gr.detach().clone()))
    recs.append(rec)
    return z
AG.conv_block = cb
pm.AG.conv_block = cb
with torch.enable_grad():
    est, pred = m(x.cuda(), train=True)
    nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(c.cuda()), c.cuda(), est)[0].backward()
AG.conv_block = orig_cb
pm.AG.conv_block = orig_cb
sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / (b.detach().cpu().double().norm() + 1e-30))
blocks = list(m.std_DCCRN.encoders) + list(m.std_DCCRN.decoders)
for i, rec in enumerate(recs):
    tr = rec["conv"]._transposed
    name = f"{'decoders' if tr else 'encoders'}.{i - 6 if tr else i}"
    dz5 = ops.rewrap(rec["dz"], rec["z"]).tensor5().cpu()
    # HIP block alone, same inputs, same upstream gradient
    xp = ops.Planar(rec["x"].buf.detach().clone().requires_grad_(True), *AG._geom(rec["x"]))
    sp = ops.Planar(rec["skip"].buf.detach().clone().requires_grad_(True), *AG._geom(rec["skip"])) if rec["skip"] is not None else None
    blk = blocks[i]
    blk.bn.init_flag = True
    with torch.enable_grad():
        z = AG.conv_block(rec["conv"], blk.bn, blk.prelu.weight, xp, sp, False)
        z.buf.backward(rec["dz"])
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = rec["x"].tensor5().detach().cpu().to(dt).clone().requires_grad_(True)
        ss = rec["skip"].tensor5().detach().cpu().to(dt).clone().requires_grad_(True) if rec["skip"] is not None else None
        sdd = {k: v.to(dt) if v.dtype.is_floating_point else v for k, v in sd.items()}
        xin = xx if ss is None else torch.cat([xx, ss], 1)
        fn = O.decoder_block if tr else O.encoder_block
        y = fn(xin, sdd, f"std_DCCRN.{name}.", np_, (i - 6 if tr else i), True, True, None)
        y.backward(dz5.to(dt))
        res[dt] = (xx.grad, ss.grad if ss is not None else None, y)
    gx = ops.rewrap(xp.buf.grad, xp).tensor5()
    line = f"{name}: fwd HIP {rel(z.tensor5(), res[torch.float64][2]):.1e} | dx HIP {rel(gx, res[torch.float64][0]):.1e} f32 {rel(res[torch.float32][0], res[torch.float64][0]):.1e}"
    if sp is not None:
        line += f" | dskip HIP {rel(ops.rewrap(sp.buf.grad, sp).tensor5(), res[torch.float64][1]):.1e} f32 {rel(res[torch.float32][1], res[torch.float64][1]):.1e}"
    dzn = dz5.double()
    line += f" | |dz| max/rms {float(dzn.abs().max() / dzn.pow(2).mean().sqrt()):.1e}"
    print(line)
